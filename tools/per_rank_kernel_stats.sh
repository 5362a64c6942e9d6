#!/bin/bash
# kernel trace of ONE rank's decode step at the shard shapes of the 8-GPU layouts (no exchange): what a 3.6-ms / 4.1-ms per-rank step is made of
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/per_rank
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for name in tp4dp2 dp8; do
  if [ $name = tp4dp2 ]; then F="--emulate-tp 4 --emulate-dp 2"; else F="--emulate-dp 8"; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$name -- python $R/bench.py $F --no-cpu-baseline --no-prefill --no-engine --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3 > $O/$name.log 2>&1
  grep '^{' $O/$name.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('# $name:', d['ms_per_step'], 'ms per step under the profiler,', d['config']['workload'][:60])" > $O/kernel_stats_$name.txt
  python $R/tools/rocpd_summary.py $(find $O/prof_$name -name "*.db" | head -1) >> $O/kernel_stats_$name.txt 2>&1
  rm -rf $O/prof_$name
  head -22 $O/kernel_stats_$name.txt | cut -c1-170
done
