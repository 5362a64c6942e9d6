#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s24
O=gpurun_out/s24
B2="timeout 600 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-prefill"
{
  echo "# 2 ranks sharing the GPU over gloo, one-shot failure injected after the first timed run: falls back to the group's all-reduce"
  XLLM_MI355_BENCH_INJECT_ONESHOT_FAILURE=1 $B2 --no-layouts 2>$O/inject.err | grep metric
  grep "bench\]" $O/inject.err | head -5
  echo "# no injection"
  $B2 --no-layouts 2>$O/plain.err | grep metric
} | cut -c1-1800 | tee $O/fallback.txt
