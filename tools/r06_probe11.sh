#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for rep in 1 2; do
for lib in tuning pf32prio1 pf32prio3; do
  echo "## lib=$lib PREFILL_M32=1"
  XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_$lib.so XLLM_MI355_PREFILL_M32=1 python tools/prefill_attn_one.py 2>&1 | grep "^\[prefill"
done
done
