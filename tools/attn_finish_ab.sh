#!/bin/bash
# A/B of the split-KV finishing launch (merge + int8 quantise in ONE launch, XLLM_MI355_ATTN_FINISH) on the per-rank decode steps
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_finish
mkdir -p $O
F="--no-cpu-baseline --no-prefill --no-engine --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3"
for rep in 1 2; do
for fin in 0 1; do
  for name in tp4dp2 dp8; do
    if [ $name = tp4dp2 ]; then E="--emulate-tp 4 --emulate-dp 2"; else E="--emulate-dp 8"; fi
    XLLM_MI355_ATTN_FINISH=$fin python $R/bench.py $E $F 2>/dev/null | grep '^{' | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finish=$fin $name rep $rep: %.4f ms per step' % d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
done
cd $R && XLLM_MI355_ATTN_FINISH=1 python -m pytest tests/test_gpu_parity.py -x -q -k "finish or attention_int8 or decode" -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/ab.txt
