#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s31
O=gpurun_out/s31
{
for M in 256 512; do
  GEMM_PACKED=1 python tools/gemm_bench.py $M
  GEMM_PACKED=1 XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_ws_AUX0.so python tools/gemm_bench.py $M
done
GEMM_FUSED=1 GEMM_PACKED=1 python tools/gemm_bench.py 256
GEMM_FUSED=1 GEMM_PACKED=1 XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_ws_AUX0.so python tools/gemm_bench.py 256
} 2>&1 | grep "gemm int8" | tee $O/aux.txt
