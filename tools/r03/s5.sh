#!/bin/bash
# round-3 GPU session 5: phase-staggered eight-wave packed GEMM: parity, then timing against the in-phase arm and the row-major kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s5
O=gpurun_out/s5
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
{
  for M in 256 192 512; do
    python tools/gemm_bench.py $M int8
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py $M int8
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=0 python tools/gemm_bench.py $M int8
  done
  GEMM_DIST=gauss GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py 256 int8
  for ng in 3 4 5; do GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_NG=$ng python tools/gemm_bench.py 256 int8; done
  for sl in 1 2 3 4 6 8; do GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_SLICES=$sl python tools/gemm_bench.py 256 int8; done
  GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py 256 fp8
  python tools/gemm_bench.py 256 fp8
} > $O/gemm.txt 2>&1
grep "gemm" $O/gemm.txt
