#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/r06_probe10.txt
: > $O
export XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_tuning.so
for m in 0 1; do
  for pm in 1 2; do
    echo "## PREFILL_M32=$m PREFILL_P=$pm" >> $O
    XLLM_MI355_PREFILL_M32=$m XLLM_MI355_PREFILL_P=$pm python tools/prefill_attn_one.py 2>&1 | grep "^\[prefill" >> $O
  done
done
echo "## parity, PREFILL_M32=1 (tuning library)" >> $O
XLLM_MI355_PREFILL_M32=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "prefill or chunked" -p no:cacheprovider 2>&1 | tail -15 >> $O
echo "## parity, PREFILL_M32=1 PREFILL_P=2" >> $O
XLLM_MI355_PREFILL_M32=1 XLLM_MI355_PREFILL_P=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "prefill or chunked" -p no:cacheprovider 2>&1 | tail -8 >> $O
cat $O
