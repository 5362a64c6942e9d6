#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s21
O=$R/gpurun_out/s21
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gate_up" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
for i in 1 2; do
python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-400 | tee -a $O/cfg2.txt
XLLM_MI355_GATE_UP_FUSION=0 python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-400 | tee -a $O/cfg2.txt
done
