#!/bin/bash
# round-2 closing run: bench lines, kernel traces, PMC traffic of the decode attention launch, layouts, whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O; rm -f $O/*
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-engine > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config cfg4-slice > $O/bench_cfg4_slice.json 2> $O/bench_cfg4_slice.err
timeout 300 python bench.py --config cfg5-slice > $O/bench_cfg5_slice.json 2> $O/bench_cfg5_slice.err
for K in 2 4 8; do timeout 300 python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>/dev/null; done
for K in 2 4; do timeout 300 python bench.py --emulate-tp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_tp.txt 2>/dev/null; done
prof() {  # name, bench args
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- python $R/bench.py "$@" > /dev/null 2>&1 )
  python tools/rocpd_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) 2>&1 | grep -v "at::native::\(vectorized\|elementwise\|unrolled\)" | head -45 > $O/${name}_kernel_stats.txt
}
prof bench --steps 5 --warmup 2 --no-cpu-baseline --no-engine
prof dp8 --emulate-dp 8 --steps 5 --warmup 2 --no-cpu-baseline --no-prefill --no-engine
prof cfg2 --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-engine
prof cfg4-slice --config cfg4-slice --steps 5 --warmup 2
prof cfg5-slice --config cfg5-slice --steps 5 --warmup 2
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o t -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-prefill --no-cpu-baseline --no-engine > /dev/null 2>&1 )
  n=$(echo $C | tr A-Z a-z)
  python tools/rocpd_summary.py $(find /tmp/pmc_$C -name "*.db" | head -1) --pmc 2>&1 | grep -v "at::native" | grep "paged_decode\|kernel-trace\|^#\|calls" | cut -c1-400 | head -14 > $O/pmc_$n.txt
done
timeout 1700 python -m pytest tests -q -m gpu --timeout 600 --tb=short 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -25 > $O/pytest.txt
cp gpurun_out/model_parity.jsonl $O/ 2>/dev/null
echo done > $O/done
