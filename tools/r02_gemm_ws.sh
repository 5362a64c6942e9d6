#!/bin/bash
# weight-stream decode GEMM: parity, then timings (launches replayed from a HIP graph) and ablations
export TMPDIR=/tmp
O=gpurun_out/r02ws; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "packed or pack_weight" 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python tools/gemm_bench.py 32,64,128,256 int8 2>&1 | grep -v amdgpu.ids > $O/gemm_rowmajor.txt
GEMM_PACKED=1 timeout 300 python tools/gemm_bench.py 32,64,128,256 int8 2>&1 | grep -v amdgpu.ids > $O/gemm_packed.txt
for v in NOADMA NOWDMA NOCOMPUTE NOMFMA; do
  XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_ws_$v.so GEMM_PACKED=1 timeout 200 python tools/gemm_bench.py 32,256 int8 2>&1 | grep -v amdgpu.ids | sed "s/^/[$v] /" >> $O/gemm_ablate.txt
done
for SL in 1 2 4 8; do
  GEMM_PACKED=1 XLLM_MI355_WS_SLICES=$SL timeout 120 python tools/gemm_bench.py 256,32 int8 2>&1 | grep -v amdgpu.ids >> $O/gemm_sweep_slices.txt
done
if [ "$1" != "quick" ]; then
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine > $O/bench.txt 2>&1
for K in 2 4 8; do
  timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>&1
done
fi
echo done > $O/done
