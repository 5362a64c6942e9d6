#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
O=$R/gpurun_out/s11
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-prefill --no-engine --steps 10 --warmup 3 > $O/bench.log 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db > $O/kernel_stats.txt 2>&1
head -40 $O/kernel_stats.txt
grep ms_per_step $O/bench.log | cut -c1-300
rm -rf $O/prof
