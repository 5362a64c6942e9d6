#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s17
O=gpurun_out/s17
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_16bit or mla_prefill_and_latent or matmul" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
XLLM_MI355_MLA_PREFILL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mla" -p no:cacheprovider 2>&1 | tail -3
{
  for M in 64 128 256; do
    python tools/gemm_bench.py $M bf16
    GEMM_PACKED=1 XLLM_MI355_PACKED_16=1 python tools/gemm_bench.py $M bf16
  done
  GEMM_PACKED=1 XLLM_MI355_PACKED_16=1 python tools/gemm_bench.py 32,16 bf16
  python tools/gemm_bench.py 32,16 bf16
} 2>&1 | grep gemm | tee $O/gemm16.txt
python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-600 | tee $O/cfg2.txt
XLLM_MI355_PACKED_16=0 python bench.py --config cfg2 --no-cpu-baseline --no-engine 2>&1 | grep metric | cut -c1-300 | tee -a $O/cfg2.txt
python bench.py --config cfg4-slice 2>&1 | grep metric | cut -c1-2500 | tee $O/cfg4.txt
