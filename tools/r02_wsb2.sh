#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02wsb2; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 --tb=short -x -k "matmul or group_gemm or model_prefill_chunked" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -15 > $O/pytest_wsb.txt
for M in 16 32 64; do timeout 200 python tools/gemm_bench.py $M bf16 2>/dev/null | grep gemm >> $O/gemm_bf16_wsb.txt; done
for S in 1 2 3 4; do XLLM_MI355_WSB_SLICES=$S timeout 200 python tools/gemm_bench.py 64 bf16 2>/dev/null | grep gemm | grep -v "gate_up\|lm_head" >> $O/gemm_bf16_slices.txt; done
for S in 1 2 3 4; do XLLM_MI355_WSB_SLICES=$S timeout 200 python tools/gemm_bench.py 32 bf16 2>/dev/null | grep gemm | grep -v "gate_up\|lm_head" >> $O/gemm_bf16_slices32.txt; done
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-engine > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config cfg4-slice > $O/bench_cfg4_slice.json 2> $O/bench_cfg4_slice.err
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -q -m gpu --timeout 600 --tb=short -k "variants or selector or dual or model_prefill" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -12 > $O/pytest_variants.txt
echo done > $O/done
