#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s12
O=gpurun_out/s12
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_oneshot_allreduce.py tests/test_gpu_parity.py -x -q -k "oneshot or greedy_argmax or packed_fp8" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-prefill --no-engine --steps 20 --warmup 5"
{
  echo "# single GPU"; $B
  echo "# 2 ranks sharing the GPU over gloo: auto layout (tp + one-shot), layouts block"; timeout 600 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-prefill --steps 5 --warmup 2
  echo "# the same, RCCL-less fallback (--no-oneshot-allreduce)"; timeout 600 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-prefill --steps 5 --warmup 2 --no-oneshot-allreduce --no-layouts
} 2>&1 | grep -v amdgpu.ids | grep "^#\|ms_per_step\|rror\|\[bench\]" | cut -c1-1500 | tee $O/bench.txt
