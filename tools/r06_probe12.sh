#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/r06_probe12.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_model_parity.py -x -q -k "prefill or chunked or model or golden or mla" -p no:cacheprovider 2>&1 | tail -8 >> $O
XLLM_MI355_PREFILL_P=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "prefill or chunked" -p no:cacheprovider 2>&1 | tail -4 >> $O
python tools/prefill_attn_one.py 2>&1 | grep "^\[prefill" >> $O
XLLM_MI355_PREFILL_P=2 python tools/prefill_attn_one.py 2>&1 | grep "^\[prefill" >> $O
cat $O
