#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s8
O=gpurun_out/s8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed or rope_cache or add_norm" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
{
  for M in 256 128 32; do
    python tools/gemm_bench.py $M int8
    GEMM_PACKED=1 XLLM_MI355_PACKED=1 python tools/gemm_bench.py $M int8
  done
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=2 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS8_STAGGER=0 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED=1 XLLM_MI355_WS_WAVES=4 python tools/gemm_bench.py 256 int8
  GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py 256,128 fp8
  GEMM_SHAPES=dsv3 GEMM_PACKED=1 XLLM_MI355_PACKED_FP8=1 python tools/gemm_bench.py 128 fp8
} 2>&1 | grep gemm | tee $O/gemm.txt
export XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_ws8t.so
{
python tools/ws8_timing.py 256 37888 3584
python tools/ws8_timing.py 256 3584 18944
python tools/ws8_timing.py 256 4608 3584
} 2>&1 | grep ws8t | tee $O/ws8_timing.txt
