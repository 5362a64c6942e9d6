#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02fail2; mkdir -p $O; rm -f $O/*
timeout 300 python tools/fp8_fullsize_diag.py > $O/fp8_diag.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_properties.py tests/test_shim.py tests/test_golden_fixtures.py -q -m gpu --timeout 600 --tb=short -k "chunked_prefill_bottom_right or random_geometry or dual_micro or moe or sample or fixtures or shim" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -40 > $O/pytest.txt
timeout 300 python bench.py --config cfg5-slice > $O/bench_cfg5_slice.json 2> $O/bench_cfg5_slice.err
echo done > $O/done
