#!/bin/bash
# kernel trace of the decode step through the C++ shim in the reference's operator order (bench.py --via-shim)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/via_shim
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --via-shim --no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc --no-per-rank --no-allocator-pages --steps 10 --warmup 3 > $O/bench.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
grep '^{' $O/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('# headline', d['ms_per_step'], 'via shim', d.get('via_shim'))"
grep -v "at::native" $O/kernel_stats.txt | head -40 | cut -c1-175
