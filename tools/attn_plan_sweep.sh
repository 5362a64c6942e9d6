#!/bin/bash
# launch-plan sweep of the decode-attention kernel on the per-rank shapes of the multi-GPU layouts (TP4 x DP2: B = 128 x 1 kv head;
# DP 8 replica: B = 32 x 4 kv heads; cfg2: B = 64 x 4 at ctx 2048), plain and with the int8 epilogue, over grid-level splits x heads
# per workgroup. Uses the -DXM_TUNING flavour of the library (plan overrides). Output: gpurun_out/attn_plan_sweep.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/attn_plan_sweep.txt
: > $O
export XLLM_MI355_LIB=$GRAFT_REPO_ROOT/xllm_amd/lib/libxllm_mi355_tuning.so
CASES=${1:-tp4dp2,dp8,cfg2,dp4}
for int8 in 0 1 2; do
  for hpw in 0 1 2 4; do
    for sp in 0 1 2 3 4 8; do
      env ATTN_INT8=$(( int8 > 0 )) XLLM_MI355_ATTN_FINISH=$(( int8 == 2 )) ATTN_N=40 $( [ $hpw != 0 ] && echo XLLM_MI355_DECODE_HPW=$hpw ) $( [ $sp != 0 ] && echo XLLM_MI355_DECODE_SPLITS=$sp ) \
        python tools/attn_bench.py $CASES 2>&1 | grep "^\[attn\]" | sed "s/LIB=[^ ]* //; s/^\[attn\]/[int8=$int8 hpw=$hpw sp=$sp]/" >> $O
    done
  done
done
sort -t: -k2 -n $O | head -0
python - <<'PY'
import re, collections
best = collections.defaultdict(list)
for line in open("gpurun_out/attn_plan_sweep.txt"):
    m = re.match(r"\[int8=(\d) hpw=(\d) sp=(\d)\].*?(\w+)\s+B=(\d+).*?:\s+([\d.]+) us", line)
    if m:
        best[(m.group(4), m.group(1))].append((float(m.group(6)), f"hpw={m.group(2)} sp={m.group(3)}"))
for k, v in sorted(best.items()):
    v.sort()
    print(k, "planner:", [x for x in v if x[1] == "hpw=0 sp=0"], "best 4:", v[:4])
PY
