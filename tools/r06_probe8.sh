#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_p8itime.so python tools/p8i_timing.py 2>&1 | grep "^\[p8i" | tee gpurun_out/r06_probe8.txt
