#!/bin/bash
# the round's evidence: the default bench line (roofline.traffic measured live), its kernel trace, the other configs
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-600 $O/bench.json
python bench.py --config cfg2 --no-cpu-baseline 2>/dev/null | grep metric > $O/bench_cfg2.json
python bench.py --config cfg4-slice 2>/dev/null | grep metric > $O/bench_cfg4_slice.json
python bench.py --config cfg5-slice 2>/dev/null | grep metric > $O/bench_cfg5_slice.json
# (the per-rank steps of the 8-GPU layouts -- TP4 x DP2 rank, DP8 replica -- are inside the default line since round 5: per_rank_emulated)
python bench.py --via-shim --no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc --no-per-rank 2>/dev/null | grep metric > $O/bench_via_shim.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-engine --no-gemm --no-pmc --no-per-rank --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
head -14 $O/kernel_stats.txt | cut -c1-170
for f in cfg2 cfg4_slice cfg5_slice via_shim; do python -c "
import json,sys
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d.get('ms_per_step'), d.get('value'), (d.get('roofline') or {}).get('frac'))"; done
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('attention_path_frac_in_graph'))
print('prefill', d.get('prefill')); print('per_rank', d.get('per_rank_emulated')); print('layer', d['gemm'].get('decode_layer_M256_nonattention'))
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('sample'))"
