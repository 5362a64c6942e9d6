#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02diag; mkdir -p $O; rm -f $O/*
DIAG_B=512 timeout 300 python tools/dual_diag.py > $O/dual512.txt 2>&1
DIAG_B=256 timeout 300 python tools/dual_diag.py > $O/dual256.txt 2>&1
DIAG_B=256 XLLM_MI355_DECODE_HPW=4 timeout 300 python tools/dual_diag.py > $O/dual256_hpw4.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_properties.py -q -m gpu --timeout 600 --tb=short -k "moe or mla" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -15 > $O/pytest.txt
timeout 300 python bench.py --config cfg5-slice > $O/bench_cfg5_slice.json 2> $O/bench_cfg5_slice.err
echo done > $O/done
