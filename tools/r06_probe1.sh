#!/bin/bash
# round 6, first GPU call: per-rank steps (the judge's Next #1) with / without the dual micro-batch executor
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_probe1.txt
: > $O
COMMON="--no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc --no-per-rank --no-allocator-pages --steps 20 --warmup 3"
for flags in "--emulate-tp 4 --emulate-dp 2" "--emulate-tp 4 --emulate-dp 2 --dual" "--emulate-dp 8" "--emulate-dp 8 --dual"; do
  echo "### bench.py $flags" >> $O
  timeout 400 python bench.py $flags $COMMON 2>>gpurun_out/r06_probe1.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'B': d['config']['per_gpu_batch'], 'attn_us': r['avg_launch_ms']*1e3, 'attn_graph_us': (r.get('attention_path_ms_in_graph') or 0)*1e3, 'micro_batches': d['config']['micro_batches']}))" >> $O 2>&1
done
cat $O
