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
python bench.py --emulate-dp 8 --no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc 2>/dev/null | grep metric > $O/bench_dp8.json
python bench.py --emulate-tp 4 --emulate-dp 2 --no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc 2>/dev/null | grep metric > $O/bench_tp4dp2_shapes.json
python bench.py --via-shim --no-cpu-baseline --no-prefill --no-gemm --no-engine --no-pmc 2>/dev/null | grep metric > $O/bench_via_shim.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-engine --no-gemm --no-pmc --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
head -14 $O/kernel_stats.txt | cut -c1-170
for f in cfg2 cfg4_slice cfg5_slice dp8 tp4dp2_shapes via_shim; do python -c "
import json,sys
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d.get('ms_per_step'), d.get('value'), (d.get('roofline') or {}).get('frac'))"; done
