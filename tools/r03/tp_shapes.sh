#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s32
O=$R/gpurun_out/s32
B="python bench.py --no-cpu-baseline --no-engine --no-gemm --no-prefill --steps 20 --warmup 5"
for k in 2 4; do
  echo "# emulate-tp $k"; $B --emulate-tp $k 2>&1 | grep ms_per_step | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ms_per_step', d['ms_per_step'], d['config']['parallelism'], d['roofline']['avg_launch_ms'])"
done | tee $O/tp.txt
echo "# emulate-tp 4, B=128 (tp4dp2 replica)" | tee -a $O/tp.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-engine --no-gemm --no-prefill --steps 10 --warmup 3 --emulate-tp 4 > $O/bench_tp4.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) | grep "xm::" | head -24 | cut -c1-100,112-170 | tee $O/tp4_kernels.txt
rm -rf $O/prof
