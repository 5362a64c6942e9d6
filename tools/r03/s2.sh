#!/bin/bash
# round-3 GPU session 2: K-walk stagger A/B on the decode GEMMs, attention epilogue A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
O=gpurun_out/s2
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "packed or int8_gemm or scaled_matmul" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
{
  for st in 1 0; do
    for M in 256 128 64 32; do
      XLLM_MI355_KSTAGGER=$st python tools/gemm_bench.py $M int8
      XLLM_MI355_KSTAGGER=$st GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py $M int8
    done
    XLLM_MI355_KSTAGGER=$st GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_WAVES=4 python tools/gemm_bench.py 256 int8
  done
  for ng in 3 4 5; do
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_NG=$ng python tools/gemm_bench.py 256 int8
  done
  for sl in 1 2 3 4 6; do
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_SLICES=$sl python tools/gemm_bench.py 256 int8
  done
} > $O/gemm_int8.txt 2>&1
grep "gemm" $O/gemm_int8.txt
{
  for st in 1 0; do
    XLLM_MI355_KSTAGGER=$st GEMM_SHAPES=dsv3 GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py 128 fp8
    XLLM_MI355_KSTAGGER=$st GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py 256 fp8
  done
} > $O/gemm_fp8.txt 2>&1
grep "gemm" $O/gemm_fp8.txt
python tools/attn_ab.py > $O/attn_ab.txt 2>&1
grep attn_ab $O/attn_ab.txt
