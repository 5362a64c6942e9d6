#!/bin/bash
# round-3 GPU session 1: eight-wave packed int8 GEMM + packed fp8 GEMM (parity, then timing), decode-attention in-situ diagnosis
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s1
O=gpurun_out/s1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed" > $O/pytest_packed.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_packed.txt
tail -5 $O/pytest_packed.txt
{
  echo "# int8, Qwen2-7B layer shapes, graph replay, us per launch"
  for M in 256 128 32; do
    python tools/gemm_bench.py $M int8
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py $M int8
  done
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_WAVES=4 python tools/gemm_bench.py 256 int8
  GEMM_DIST=gauss python tools/gemm_bench.py 256 int8
  GEMM_DIST=gauss GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
  for ng in 3 4 5; do
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_NG=$ng python tools/gemm_bench.py 256 int8
  done
  for sl in 1 2 4; do
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_SLICES=$sl python tools/gemm_bench.py 256 int8
  done
} > $O/gemm_int8.txt 2>&1
{
  echo "# fp8, DeepSeek-V3 MLA projections of one TP=8 rank and Qwen2-7B shapes"
  for M in 128 32; do
    GEMM_SHAPES=dsv3 python tools/gemm_bench.py $M fp8
    GEMM_SHAPES=dsv3 GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py $M fp8
  done
  for M in 128 256; do
    python tools/gemm_bench.py $M fp8
    GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py $M fp8
  done
} > $O/gemm_fp8.txt 2>&1
{
  echo "# decode attention tp1 (cfg3): 3 vs 28 rotating KV copies, plain vs int8-epilogue"
  python tools/attn_bench.py tp1
  ATTN_INT8=1 python tools/attn_bench.py tp1
  ATTN_COPIES=28 ATTN_N=56 python tools/attn_bench.py tp1
  ATTN_COPIES=28 ATTN_N=56 ATTN_INT8=1 python tools/attn_bench.py tp1
} > $O/attn_insitu.txt 2>&1
cat $O/gemm_int8.txt | grep -v "^$" | tail -60
cat $O/gemm_fp8.txt | tail -40
cat $O/attn_insitu.txt
