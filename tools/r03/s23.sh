#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s23
O=$R/gpurun_out/s23
export GEMM_SHAPES_ONLY=gate_up
{
for M in 256 32; do
  GEMM_PACKED=1 python tools/gemm_bench.py $M
  GEMM_GU=2 python tools/gemm_bench.py $M
  GEMM_GU=1 python tools/gemm_bench.py $M
  GEMM_GU=1 XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_gu_NOATOMIC.so python tools/gemm_bench.py $M
  GEMM_GU=1 XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_gu_NOLDSATOMIC.so python tools/gemm_bench.py $M
done
} 2>&1 | grep "gate_up" | tee $O/gu.txt
