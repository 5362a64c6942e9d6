#!/bin/bash
# A/B of the grouped decode GEMM's plan knobs through the tuning library: rows per pass (16 / 32) x per-wave column tiles (short K)
T=$GRAFT_REPO_ROOT/xllm_amd/lib/libxllm_mi355_tuning.so
for skew in 0 1; do
for mb in 1 2; do
for wc in 0 1; do
  echo "## skew=$skew rows-per-pass=$((16*mb)) wave-cols=$wc"
  GG_SKEW=$skew XLLM_MI355_LIB=$T XLLM_MI355_WSB_GROUP_MB=$mb XLLM_MI355_WSB_WAVE_COLS=$wc python tools/group_gemm_decode_bench.py 2>&1 | grep "group gemm" | cut -c1-150
done; done; done
