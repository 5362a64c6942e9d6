#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02small; mkdir -p $O; rm -f $O/*
timeout 300 python bench.py --config cfg5-slice > $O/bench_cfg5_slice.json 2> $O/bench_cfg5_slice.err
timeout 300 python -m pytest tests/test_forward_input.py -q -m gpu --tb=short 2>&1 | tail -5 > $O/pytest_fi.txt
timeout 600 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-prefill --no-cpu-baseline > $O/bench_dp2_gloo.json 2> $O/bench_dp2_gloo.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg5-slice --steps 5 --warmup 2 > /dev/null 2>&1 )
python tools/rocpd_summary.py $(find /tmp/prof_c5 -name "*.db" | head -1) 2>&1 | grep -v "at::native::\(vectorized\|elementwise\|unrolled\)" | head -40 > $O/cfg5-slice_kernel_stats.txt
echo done > $O/done
