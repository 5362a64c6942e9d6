#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_probe9.txt
: > $O
timeout 900 python -m pytest tests/test_shim.py -x -q -p no:cacheprovider 2>&1 | tail -25 >> $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_copy or mla" -p no:cacheprovider 2>&1 | tail -5 >> $O
cat $O
