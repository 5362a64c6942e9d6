#!/bin/bash
# whole GPU suite (all failures listed), decode-attention split sweep at the small-batch shapes, kernel traces of the two slices
export TMPDIR=/tmp
O=gpurun_out/r02verify2; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -40 > $O/pytest.txt
for S in 1 2 3 4 6 8; do
  XLLM_MI355_DECODE_SPLITS=$S timeout 120 python tools/attn_bench.py cfg2,dp2,dp4,dp8 2>/dev/null | grep attn >> $O/attn_splits.txt
done
timeout 120 python tools/attn_bench.py cfg2,dp2,dp4,dp8,tp1 2>/dev/null | grep attn >> $O/attn_splits.txt
for C in cfg4-slice cfg5-slice; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$C -o s -- python $GRAFT_REPO_ROOT/bench.py --config $C --steps 5 --warmup 2 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_summary.py $(find /tmp/prof_$C -name "*.db" | head -1) 2>&1 | head -30 > $O/${C}_kernel_stats.txt
done
echo done > $O/done
