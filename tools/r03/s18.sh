#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s18
O=gpurun_out/s18
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gate_up or packed_gemm" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-engine --no-gemm --steps 20 --warmup 5"
{
  echo "# fused gate_up"; $B
  echo "# XLLM_MI355_GATE_UP_FUSION=0"; XLLM_MI355_GATE_UP_FUSION=0 $B
  echo "# dp8 fused"; $B --emulate-dp 8 --no-prefill
  echo "# dp8 unfused"; XLLM_MI355_GATE_UP_FUSION=0 $B --emulate-dp 8 --no-prefill
} 2>&1 | grep -v amdgpu.ids | grep "^#\|ms_per_step" | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('#'): print(line.strip()); continue
    d=json.loads(line); print(' ms_per_step', d['ms_per_step'], 'prefill', d.get('prefill'))
" | tee $O/bench.txt
