#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s6
export XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_ws8t.so
{
python tools/ws8_timing.py 256 37888 3584
python tools/ws8_timing.py 256 3584 18944
python tools/ws8_timing.py 256 4608 3584
python tools/ws8_timing.py 512 37888 3584
} 2>&1 | grep ws8t | tee gpurun_out/s6/ws8_timing.txt
