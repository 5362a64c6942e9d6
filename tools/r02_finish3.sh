#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02finish3; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 --tb=short -k "fusion or model_step or dual or engine or qkv_gemm" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -8 > $O/pytest.txt
for K in 2 4 8; do timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>/dev/null; done
XLLM_MI355_QKV_ROPE=0 timeout 300 python bench.py --emulate-dp 8 --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp_nofusion.txt 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dp8 -o t -- python $GRAFT_REPO_ROOT/bench.py --emulate-dp 8 --steps 5 --warmup 2 --no-cpu-baseline --no-prefill --no-engine > /dev/null 2>&1 )
python tools/rocpd_summary.py $(find /tmp/prof_dp8 -name "*.db" | head -1) 2>&1 | grep -v "at::native::\(vectorized\|elementwise\|unrolled\)" | head -30 > $O/dp8_kernel_stats.txt
echo done > $O/done
