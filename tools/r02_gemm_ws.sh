#!/bin/bash
# weight-stream decode GEMM: parity, then timings against the row-major kernels, then the whole decode step
export TMPDIR=/tmp
O=gpurun_out/r02ws; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "packed or pack_weight" 2>&1 | tail -15 > $O/pytest.txt
GEMM_PACKED=1 timeout 300 python tools/gemm_bench.py 32,64,128,256 int8 > $O/gemm_packed.txt 2>&1
for NG in 2 4 6 8 10; do
  GEMM_PACKED=1 XLLM_MI355_WS_NG=$NG timeout 120 python tools/gemm_bench.py 256 int8 >> $O/gemm_sweep256.txt 2>&1
done
for SL in 1 2 3 4 6; do
  GEMM_PACKED=1 XLLM_MI355_WS_SLICES=$SL timeout 120 python tools/gemm_bench.py 256,64 int8 >> $O/gemm_sweep_slices.txt 2>&1
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine > $O/bench.txt 2>&1
for K in 2 4 8; do
  timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>&1
done
echo done > $O/done
