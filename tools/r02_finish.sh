#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02finish; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_parity.py -q -m gpu --timeout 300 --tb=short -k "decode or fusion or model_step or dual or engine or teacher" 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -15 > $O/pytest.txt
for K in 2 4 8; do timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>/dev/null; done
for K in 4 8; do XLLM_MI355_ATTN_FINISH=0 timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp_nofinish.txt 2>/dev/null; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine > $O/bench.json 2>/dev/null
echo done > $O/done
