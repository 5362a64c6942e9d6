R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/engine_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-prefill --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 10 --warmup 3 > $O/bench.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
grep -E "top_k|softmax_random|greedy|philox|random|gemm_ws8s_kernel<2|copy" $O/kernel_stats.txt | cut -c1-180
grep '^{' $O/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['engine'])"
