#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s20
O=$R/gpurun_out/s20
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gate_up" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-400 | tee $O/cfg2.txt
XLLM_MI355_GATE_UP_FUSION=0 python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-400 | tee -a $O/cfg2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-engine --no-gemm --steps 10 --warmup 3 > $O/bench.log 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db > $O/kernel_stats.txt 2>&1
head -60 $O/kernel_stats.txt | cut -c1-200
grep ms_per_step $O/bench.log | cut -c1-300
rm -rf $O/prof
