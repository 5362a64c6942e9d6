#!/bin/bash
# weight-stream 16-bit GEMM: parity, micro-benchmark against the tiled kernels, cfg2 / cfg4-slice steps, whole suite
export TMPDIR=/tmp
O=gpurun_out/r02wsb; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 --tb=short -x -k "matmul or group_gemm or moe or dual_micro" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -25 > $O/pytest_wsb.txt
for M in 16 32 64; do
  timeout 200 python tools/gemm_bench.py $M bf16 2>/dev/null | grep gemm >> $O/gemm_bf16_wsb.txt
  XLLM_MI355_WSB=0 timeout 200 python tools/gemm_bench.py $M bf16 2>/dev/null | grep gemm >> $O/gemm_bf16_tiled.txt
done
for S in 1 2 4 6 8; do XLLM_MI355_WSB_SLICES=$S timeout 200 python tools/gemm_bench.py 64 bf16 2>/dev/null | grep gemm >> $O/gemm_bf16_slices.txt; done
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-engine > $O/bench_cfg2.json 2> $O/bench_cfg2.err
XLLM_MI355_WSB=0 timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-engine > $O/bench_cfg2_tiled.json 2> /dev/null
timeout 300 python bench.py --config cfg4-slice > $O/bench_cfg4_slice.json 2> $O/bench_cfg4_slice.err
XLLM_MI355_WSB=0 timeout 300 python bench.py --config cfg4-slice > $O/bench_cfg4_slice_tiled.json 2> /dev/null
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -30 > $O/pytest.txt
echo done > $O/done
