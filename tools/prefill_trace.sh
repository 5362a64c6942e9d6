#!/bin/bash
# kernel trace of the prefill chunk (8192 tokens through the 28 layers): bench.py with ONE decode step so that the chunk's four runs dominate the trace
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pf -- python $R/bench.py --no-cpu-baseline --no-gemm --no-engine --no-pmc --no-per-rank --no-allocator-pages --steps 1 --warmup 1 > $R/gpurun_out/prefill_trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_pf -name "*.db" | head -1) > $R/gpurun_out/r06_prefill_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_pf
grep '^{' $R/gpurun_out/prefill_trace.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('# prefill', d['prefill'])"
head -30 $R/gpurun_out/r06_prefill_kernel_stats.txt | cut -c1-175
