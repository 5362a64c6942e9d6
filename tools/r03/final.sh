#!/bin/bash
# round-3 final evidence: the default bench line, its kernel trace, the PMC passes of the decode-attention launch, the other configs
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
O=$R/gpurun_out/final
python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
python bench.py --via-shim --no-cpu-baseline --no-prefill --no-gemm --no-engine 2>/dev/null | grep metric > $O/bench_via_shim.json
python bench.py --emulate-dp 8 --no-cpu-baseline --no-prefill --no-gemm --no-engine 2>/dev/null | grep metric > $O/bench_dp8.json
python bench.py --config cfg2 --no-cpu-baseline 2>/dev/null | grep metric > $O/bench_cfg2.json
python bench.py --config cfg4-slice 2>/dev/null | grep metric > $O/bench_cfg4_slice.json
python bench.py --config cfg5-slice 2>/dev/null | grep metric > $O/bench_cfg5_slice.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-engine --no-gemm --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-prefill --no-cpu-baseline --no-engine --no-gemm > $O/pmc_$C.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$C -name "*.db" | head -1) --pmc | grep "paged_decode" > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
head -12 $O/kernel_stats.txt | cut -c1-180
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt | cut -c1-60,110-250
