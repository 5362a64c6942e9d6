#!/bin/bash
# round 6: eight-wave decode workgroups (A/B against four waves through the tuning library) + parity of the decode attention tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_probe2.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode or paged or fusion or model_step" -p no:cacheprovider 2>&1 | tail -5 >> $O
export XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_tuning.so
for int8 in 0 1; do
  for w in 4 8; do
    for sp in 0 1 2; do
      env ATTN_INT8=$int8 ATTN_N=40 XLLM_MI355_DECODE_WAVES=$w $( [ $sp != 0 ] && echo XLLM_MI355_DECODE_SPLITS=$sp ) \
        python tools/attn_bench.py tp4dp2,dp8,cfg2,dp4 2>&1 | grep "^\[attn\]" | sed "s/LIB=[^ ]* //; s/^\[attn\]/[int8=$int8 waves=$w sp=$sp]/" >> $O
    done
  done
done
unset XLLM_MI355_LIB
COMMON="--no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3"
for flags in "--emulate-tp 4 --emulate-dp 2" "--emulate-dp 8"; do
  echo "### bench.py $flags" >> $O
  timeout 400 python bench.py $flags $COMMON 2>>gpurun_out/r06_probe2.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'B': d['config']['per_gpu_batch'], 'attn_us': r['avg_launch_ms']*1e3, 'attn_graph_us': (r.get('attention_path_ms_in_graph') or 0)*1e3}))" >> $O 2>&1
done
cat $O
