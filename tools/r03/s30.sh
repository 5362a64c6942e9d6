#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s30
O=gpurun_out/s30
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "qkv_gemm_rope or add_rms_norm or golden or decoder" -p no:cacheprovider 2>&1 | tail -3
python tools/step_ab.py 32 4096 base= o_1_1=shape=3584,3584,1,1 o_2_2=shape=3584,3584,2,2 down_1_4=shape=3584,18944,1,4 down_2_4=shape=3584,18944,2,4 2>&1 | grep "B=" | tee $O/ab32.txt
python tools/step_ab.py 64 4096 base= o_1_1=shape=3584,3584,1,1 down_1_4=shape=3584,18944,1,4 2>&1 | grep "B=" | tee $O/ab64.txt
python tools/step_ab.py 256 4096 base= 2>&1 | grep "B=" | tee $O/ab256.txt
