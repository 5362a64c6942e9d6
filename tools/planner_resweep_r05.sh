#!/bin/bash
# round 5: the per-shape (tile width, K slices) plans of the three sliced decode GEMMs re-swept inside the step (tools/step_ab.py, tuning flavour)
# after the kernel changes since the round-3 sweep (non-temporal slabs, compile-time stream policy, epilogue rewrite)
cd "$GRAFT_REPO_ROOT" || exit 1
export XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_tuning.so
B=${1:-256}
python tools/step_ab.py $B 4096 base= qkv_4_3=shape=4608,3584,4,3 qkv_5_3=shape=4608,3584,5,3 qkv_4_4=shape=4608,3584,4,4 qkv_5_4=shape=4608,3584,5,4 qkv_3_4=shape=4608,3584,3,4 qkv_4_2=shape=4608,3584,4,2 qkv_5_2=shape=4608,3584,5,2 qkv_3_3=shape=4608,3584,3,3 qkv_3_5=shape=4608,3584,3,5 2>&1 | grep step_ab
python tools/step_ab.py $B 4096 base= o_4_4=shape=3584,3584,4,4 o_4_3=shape=3584,3584,4,3 o_3_4=shape=3584,3584,3,4 o_5_3=shape=3584,3584,5,3 o_4_2=shape=3584,3584,4,2 o_5_4=shape=3584,3584,5,4 o_3_5=shape=3584,3584,3,5 o_2_5=shape=3584,3584,2,5 2>&1 | grep step_ab
python tools/step_ab.py $B 4096 base= down_3_4=shape=3584,18944,3,4 down_4_4=shape=3584,18944,4,4 down_3_5=shape=3584,18944,3,5 down_4_3=shape=3584,18944,4,3 down_3_6=shape=3584,18944,3,6 down_2_6=shape=3584,18944,2,6 down_3_8=shape=3584,18944,3,8 down_5_3=shape=3584,18944,5,3 2>&1 | grep step_ab
