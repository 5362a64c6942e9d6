#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02bench; mkdir -p $O; rm -f $O/*
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.txt 2>&1
timeout 600 python bench.py --gpus 2 --backend gloo --config tiny --steps 5 --warmup 2 --no-prefill > $O/bench_gloo_dp2.txt 2>&1
timeout 900 python bench.py --gpus 2 --backend gloo --layout tp --steps 3 --warmup 2 --no-prefill > $O/bench_gloo_tp2.txt 2>&1
echo done > $O/done
