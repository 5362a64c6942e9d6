#!/bin/bash
# round-2 verification pass: whole GPU suite, the bench line, the two TP=8 slices, kernel trace of the bench command
export TMPDIR=/tmp
O=gpurun_out/r02verify; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests -q -m gpu -x --timeout 600 2>&1 | tail -15 > $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --config cfg4-slice > $O/bench_cfg4_slice.json 2> $O/bench_cfg4_slice.err
timeout 300 python bench.py --config cfg5-slice > $O/bench_cfg5_slice.json 2> $O/bench_cfg5_slice.err
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-engine > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-engine > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_bench -name "*.db" | head -1) 2>&1 | grep -v "at::native" | head -40 > $O/bench_kernel_stats.txt
cp gpurun_out/model_parity.jsonl $O/ 2>/dev/null
echo done > $O/done
