#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/r06_probe6.txt
: > $O
python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
GG_ROWS_PER_EXPERT=512 python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
GG_ROWS_PER_EXPERT=256 python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gg -- python $R/tools/group_gemm_bench.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_gg -name "*.db" | head -1) 2>&1 | grep -v "at::native" | head -12 | cut -c1-180 >> $O
rm -rf $R/gpurun_out/prof_gg
cat $O
