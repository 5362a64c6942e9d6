#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s34
O=gpurun_out/s34
timeout 900 python -m pytest tests/test_gpu_oneshot_allreduce.py -x -q -p no:cacheprovider 2>&1 | tail -15
timeout 600 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-prefill --no-layouts 2>$O/tp2.err | grep metric | cut -c1-900 | tee $O/tp2.txt
grep -i "error\|bench\]" $O/tp2.err | head -5
