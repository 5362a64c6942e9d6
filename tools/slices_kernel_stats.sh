#!/bin/bash
# kernel traces of the cfg4 / cfg5 layer slices (bench.py --config cfg4-slice | cfg5-slice under rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/slices
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cfg4-slice cfg5-slice; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -- python $R/bench.py --config $c --no-pmc > $O/$c.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1
  rm -rf $O/prof_$c
  echo "== $c"; grep '^{' $O/$c.log | tail -1 | cut -c1-200
  grep -v "at::native" $O/kernel_stats_$c.txt | head -34 | cut -c1-175
done
