#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s15
{
  echo "# planner's own choice"
  GEMM_FUSED=1 GEMM_PACKED=1 python tools/gemm_bench.py 256 int8
  for ng in 1 2 3 4 5; do for sl in 1 2 3 4 6 8; do
    GEMM_FUSED=1 GEMM_PACKED=1 XLLM_MI355_WS_NG=$ng XLLM_MI355_WS_SLICES=$sl GEMM_N=50 GEMM_REPLAYS=6 python tools/gemm_bench.py 256 int8
  done; done
} 2>&1 | grep "gemm\|^#" | grep -v gate_up | tee gpurun_out/s15/sweep.txt
