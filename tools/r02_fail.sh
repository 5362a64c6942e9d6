#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02fail; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_properties.py -q -m gpu --timeout 600 --tb=short -k "fp8_gemm_full_size or chunked_prefill_bottom_right or random_geometry or test_paged or decode" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -150 > $O/pytest.txt
timeout 120 python tools/attn_bench.py cfg2,dp2,dp4,dp8,tp1 2>/dev/null | grep attn >> $O/attn_hpw.txt
for H in 1 2; do for S in 1 2 3 4; do
  XLLM_MI355_DECODE_HPW=$H XLLM_MI355_DECODE_SPLITS=$S timeout 120 python tools/attn_bench.py cfg2,dp2,dp4,dp8 2>/dev/null | grep attn >> $O/attn_hpw.txt
done; done
echo done > $O/done
