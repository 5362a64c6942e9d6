#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s13
O=gpurun_out/s13
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_model_parity.py tests/test_properties.py -x -q -k "prefill or chunked or mla or golden" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -25 $O/pytest.txt
cat gpurun_out/model_parity.jsonl 2>/dev/null | tail -2 | cut -c1-1500
