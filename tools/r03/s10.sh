#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s10
O=gpurun_out/s10
B="python bench.py --no-cpu-baseline --no-prefill --no-engine --steps 20 --warmup 5"
{
  echo "# policy r2"; XLLM_MI355_PACKED=r2 $B
  echo "# policy auto (r3)"; $B
  echo "# policy auto (r3), in-phase eight-wave"; XLLM_MI355_WS8_STAGGER=0 $B
  echo "# policy all packed"; XLLM_MI355_PACKED=1 $B
  echo "# policy all packed, in-phase"; XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=0 $B
  echo "# dp8 replica, auto"; $B --emulate-dp 8
  echo "# dp8 replica, r2"; XLLM_MI355_PACKED=r2 $B --emulate-dp 8
} 2>&1 | grep -v amdgpu.ids | grep "^#\|ms_per_step" | sed 's/"cpu_baseline.*//' | tee $O/bench.txt
