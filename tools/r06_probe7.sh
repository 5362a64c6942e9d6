#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/r06_probe7.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "moe or group" -p no:cacheprovider 2>&1 | tail -5 >> $O
python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
GG_ROWS_PER_EXPERT=512 python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
GG_ROWS_PER_EXPERT=256 python tools/group_gemm_bench.py 2>&1 | grep "^\[ggemm" >> $O
cat $O
