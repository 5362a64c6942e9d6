#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_probe4.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sample or engine or top_k" -p no:cacheprovider 2>&1 | tail -15 >> $O
timeout 600 python tools/sampler_bench.py >> $O 2>&1
cat $O
