#!/bin/bash
# sweeps the skinny-GEMM tuning knobs (env overrides read once per process) on the decode shapes
cd $(dirname $0)/..
for wv in 4 8; do for dp in 2 3; do for nt in 1 2 3 4 5; do
  XLLM_MI355_SKINNY_WAVES=$wv XLLM_MI355_SKINNY_DEPTH=$dp XLLM_MI355_SKINNY_NT=$nt python tools/gemm_bench.py ${1:-256} int8 2>&1 | grep "gemm int8" | awk -v c="wv=$wv dp=$dp nt=$nt" '{printf "%s %s %s us | ", c, $(NF-9), $(NF-5)} END{print ""}'
done; done; done
