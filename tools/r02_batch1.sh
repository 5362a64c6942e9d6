#!/bin/bash
# round-2 GPU batch 1: logits-level parity (both prefill P modes), prefill P accuracy / speed, baselines at the DP shapes
export TMPDIR=/tmp
O=gpurun_out/r02b1; mkdir -p $O
rm -f gpurun_out/model_parity.jsonl
python -m pytest tests/test_gpu_model_parity.py -q 2>&1 | tail -25 > $O/parity_p2.txt
mv gpurun_out/model_parity.jsonl $O/model_parity_p2.jsonl
XLLM_MI355_PREFILL_P=1 python -m pytest tests/test_gpu_model_parity.py -q -k "0_5b" 2>&1 | tail -25 > $O/parity_p1.txt
mv gpurun_out/model_parity.jsonl $O/model_parity_p1.jsonl
for P in 1 2; do
  XLLM_MI355_PREFILL_P=$P python tools/prefill_p_accuracy.py >> $O/prefill_p.txt 2>&1
  XLLM_MI355_PREFILL_P=$P python tools/prefill_attn_one.py >> $O/prefill_p.txt 2>&1
done
python tools/gemm_bench.py 32,64,128,256 int8 > $O/gemm_base.txt 2>&1
python tools/attn_bench.py tp1,dp2,dp4,dp8,cfg2 > $O/attn_base.txt 2>&1
for K in 2 4 8; do
  python bench.py --emulate-dp $K --steps 10 --warmup 3 --no-cpu-baseline --no-prefill --no-engine >> $O/bench_dp.txt 2>&1
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_dp8 -o dp8 -- python $GRAFT_REPO_ROOT/bench.py --emulate-dp 8 --steps 5 --warmup 2 --no-cpu-baseline --no-prefill --no-engine > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_dp8 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/dp8_kernel_stats.txt 2>&1
find /tmp/prof_dp8 -name "*stats*.csv" -exec cp {} $O/ \;
echo done > $O/done
