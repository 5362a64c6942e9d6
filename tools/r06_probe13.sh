#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
export XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_tuning.so
for rep in 1 2; do
for deep in 0 1; do
  env ATTN_N=60 XLLM_MI355_DECODE_DEEP=$deep python tools/attn_bench.py cfg2,tp4dp2,dp8,tp1 2>&1 | grep "^\[attn\]" | sed "s/LIB=[^ ]* //"
done
done
