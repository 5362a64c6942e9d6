#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s16
O=gpurun_out/s16
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
python bench.py --config cfg4-slice 2>&1 | grep -v amdgpu.ids | grep "metric" | cut -c1-2500 > $O/bench_cfg4_slice.json
python bench.py --config cfg5-slice 2>&1 | grep -v amdgpu.ids | grep "metric" | cut -c1-2500 > $O/bench_cfg5_slice.json
python bench.py --config cfg2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | grep "metric" | cut -c1-2500 > $O/bench_cfg2.json
cat $O/bench_cfg4_slice.json $O/bench_cfg5_slice.json $O/bench_cfg2.json
