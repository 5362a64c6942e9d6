#!/bin/bash
# round-3 GPU session 3: what bounds the M = 256 decode GEMMs? ablation builds of the eight-wave packed kernel + weights from the
# Infinity Cache (one weight copy re-used by every launch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
O=gpurun_out/s3
export TMPDIR=/tmp
L=$PWD/xllm_amd/lib
{
  echo "# reference: full kernel"
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256,128 int8
  for v in NOADMA NOWDMA NOCOMPUTE NOMFMA; do
    echo "# ablation $v (eight-wave tile)"
    XLLM_MI355_LIB=$L/libxllm_mi355_ws_$v.so GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256,128 int8
  done
  for v in NOADMA NOWDMA NOCOMPUTE NOMFMA; do
    echo "# ablation $v (four-wave tile)"
    XLLM_MI355_WS_WAVES=4 XLLM_MI355_LIB=$L/libxllm_mi355_ws_$v.so GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
  done
  echo "# one weight copy (Infinity Cache resident), packed and row-major"
  GEMM_COPIES=1 GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256,128,32 int8
  GEMM_COPIES=1 python tools/gemm_bench.py 256,128 int8
} > $O/gemm_ablate.txt 2>&1
grep "gemm\|^#" $O/gemm_ablate.txt
