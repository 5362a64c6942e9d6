#!/bin/bash
# pre-flight of the multi-GPU paths on a ONE-GPU box: N ranks share the GPU over gloo (functional only; times meaningless)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s36
O=gpurun_out/s36
for N in 4 8; do
  echo "# --gpus $N --backend gloo (ranks share the GPU)" | tee -a $O/multi.txt
  timeout 900 python bench.py --gpus $N --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-prefill 2>$O/n$N.err | grep metric | cut -c1-2200 | tee -a $O/multi.txt
  grep -i "error\|bench\]\|Traceback" $O/n$N.err | head -5 | tee -a $O/multi.txt
done
