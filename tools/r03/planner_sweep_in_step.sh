#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s29
O=gpurun_out/s29
python tools/step_ab.py 256 4096 base= qkv_2_1=shape=4608,3584,2,1 qkv_5_2=shape=4608,3584,5,2 qkv_3_4=shape=4608,3584,3,4 o_1_1=shape=3584,3584,1,1 o_2_2=shape=3584,3584,2,2 o_4_2=shape=3584,3584,4,2 down_2_2=shape=3584,18944,2,2 down_4_2=shape=3584,18944,4,2 down_5_4=shape=3584,18944,5,4 down_4_8=shape=3584,18944,4,8 2>&1 | grep step_ab | tee $O/ab256.txt
python tools/step_ab.py 32 4096 base= qkv_2_2=shape=4608,3584,2,2 qkv_1_1=shape=4608,3584,1,1 o_1_2=shape=3584,3584,1,2 o_2_2=shape=3584,3584,2,2 down_2_4=shape=3584,18944,2,4 down_3_8=shape=3584,18944,3,8 down_1_4=shape=3584,18944,1,4 2>&1 | grep step_ab | tee $O/ab32.txt
