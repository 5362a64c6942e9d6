#!/bin/bash
# A/B of the slab consumers (acc_add_rms_norm & co. with every load in flight) on the decode steps: new library vs XLLM_MI355_LIB=<old build>
# (the old build: `git show <commit before 90a3ea2>:xllm_amd/csrc/rowwise.hip` compiled to an object, linked with the other objects of
#  xllm_amd/csrc/build/ into xllm_amd/lib/libxllm_mi355_oldrow.so -- not kept in the tree)
R=$GRAFT_REPO_ROOT
OLD=$R/xllm_amd/lib/libxllm_mi355_oldrow.so
F="--no-cpu-baseline --no-prefill --no-engine --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3"
for rep in 1 2; do
for lib in new old; do
  for name in headline tp4dp2 dp8; do
    case $name in headline) E="";; tp4dp2) E="--emulate-tp 4 --emulate-dp 2";; dp8) E="--emulate-dp 8";; esac
    if [ $lib = old ]; then export XLLM_MI355_LIB=$OLD; else unset XLLM_MI355_LIB; fi
    python $R/bench.py $E $F 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $name rep $rep: %.4f ms per step' % d['ms_per_step'])"
  done
done
done
unset XLLM_MI355_LIB
cd $R && python -m pytest tests/test_gpu_parity.py -x -q -k "norm or fusion or slab or model_step" -p no:cacheprovider 2>&1 | tail -3
