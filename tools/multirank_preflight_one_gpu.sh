#!/bin/bash
# round 4 pre-flight of the multi-GPU paths on a ONE-GPU box: N ranks share the GPU over gloo (functional only; times meaningless).
# New this round: the fused lm_head + argmax with the [B] pair exchange under TP, and the three exchange arms of `layouts`
# (one-shot kernel = headline, <layout>_rccl, <layout>_rccl_overlap).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/multirank
mkdir -p $O
for N in ${1:-2 4}; do
  echo "# --gpus $N --backend gloo (ranks share the GPU)" | tee -a $O/multi.txt
  timeout 1500 python bench.py --gpus $N --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-prefill --time-limit-s 600 2>$O/n$N.err | grep metric | cut -c1-3000 | tee -a $O/multi.txt
  grep -i "error\|bench\]\|Traceback" $O/n$N.err | head -8 | tee -a $O/multi.txt
done
