#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s9
O=gpurun_out/s9
{
  for rep in 1 2; do
  python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=2 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=0 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_WAVES=4 python tools/gemm_bench.py 256 int8
  done
  for M in 128 64 32; do
    python tools/gemm_bench.py $M int8
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py $M int8
  done
} 2>&1 | grep gemm | tee $O/gemm.txt
