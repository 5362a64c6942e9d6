#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s14
O=gpurun_out/s14
timeout 900 python -m pytest tests/test_shim.py tests/test_gpu_parity.py -x -q -k "shim or moe or splitk or dual" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -25 $O/pytest.txt
python bench.py --no-cpu-baseline --no-prefill --steps 20 --warmup 5 --via-shim 2>&1 | grep -v amdgpu.ids | grep "ms_per_step\|rror" | cut -c1-3000 | tee $O/bench_via_shim.txt
