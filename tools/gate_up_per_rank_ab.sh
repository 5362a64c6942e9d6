R=$GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-prefill --no-engine --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3"
for gu in 1 0; do
  for name in tp4dp2 dp8; do
    if [ $name = tp4dp2 ]; then E="--emulate-tp 4 --emulate-dp 2"; else E="--emulate-dp 8"; fi
    XLLM_MI355_GATE_UP_FUSION=$gu python $R/bench.py $E $F 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gate_up_fusion=$gu $name: %.4f ms per step' % d['ms_per_step'])"
  done
done
