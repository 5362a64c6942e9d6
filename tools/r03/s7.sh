#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s7
{
GEMM_DIST=zero GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
GEMM_DIST=zero python tools/gemm_bench.py 256 int8
GEMM_DIST=gauss GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
GEMM_PACKED=1 XLLM_MI355_PACKED=1 GEMM_GRAPH=0 python tools/gemm_bench.py 256 int8
rocm-smi --showclocks --showpower 2>&1 | head -30
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s7/zero.txt
