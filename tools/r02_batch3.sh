#!/bin/bash
# one-shot all-reduce protocol test, TP bench with it on two ranks sharing the GPU, decode-attention A/B switches, short-launch HBM ceiling, whole suite
export TMPDIR=/tmp
O=gpurun_out/r02b3; mkdir -p $O; rm -f $O/*
timeout 400 python -m pytest tests/test_gpu_oneshot_allreduce.py -q -m gpu --timeout 300 --tb=short 2>&1 | grep -v "Warning\|^$" | tail -25 > $O/oneshot_pytest.txt
timeout 600 python bench.py --gpus 2 --backend gloo --layout tp --steps 5 --warmup 2 --no-prefill --no-cpu-baseline --oneshot-allreduce > $O/bench_tp2_oneshot.json 2> $O/bench_tp2_oneshot.err
timeout 600 python bench.py --gpus 2 --backend gloo --layout tp --steps 5 --warmup 2 --no-prefill --no-cpu-baseline > $O/bench_tp2_gloo.json 2> $O/bench_tp2_gloo.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_read_bench tools/hbm_read_bench.hip && timeout 120 /tmp/hbm_read_bench small > $O/hbm_short.txt 2>&1
for E in "XLLM_MI355_DECODE_EXCL=1" "XLLM_MI355_DECODE_DEEP=1" "XLLM_MI355_DECODE_EXCL=1 XLLM_MI355_DECODE_DEEP=1"; do
  env $E timeout 120 python tools/attn_bench.py cfg2,dp2,dp4,dp8,tp1 2>/dev/null | grep attn >> $O/attn_ab.txt
done
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short --deselect tests/test_gpu_oneshot_allreduce.py 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -40 > $O/pytest.txt
echo done > $O/done
