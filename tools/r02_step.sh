#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02step; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "packed or pack_weight or fusion or model_step" 2>&1 | tail -5 > $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine > $O/bench.txt 2>&1
for K in 2 4 8; do
  timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>&1
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_dp8 -o dp8 -- python $GRAFT_REPO_ROOT/bench.py --emulate-dp 8 --steps 5 --warmup 2 --no-cpu-baseline --no-prefill --no-engine > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_dp8 -name "*.db" | head -1) 2>&1 | grep -v "at::native" | head -30 > $O/dp8_kernel_stats.txt
echo done > $O/done
