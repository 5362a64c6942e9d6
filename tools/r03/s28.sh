#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s28
O=gpurun_out/s28
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -k "packed or gate_up or qkv_gemm_rope or add_rms_norm or prefill_parity_under or golden or decoder or model" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
python tools/step_ab.py 256 4096 base= rows256=ws_waves=131 scalar_rope=env:XLLM_MI355_SLAB_ROPE_VEC=0 2>&1 | grep step_ab | tee $O/ab256.txt
python tools/step_ab.py 32 4096 base= 2>&1 | grep step_ab | tee $O/ab32.txt
python tools/step_ab.py 128 4096 base= rows256=ws_waves=131 2>&1 | grep step_ab | tee $O/ab128.txt
for M in 160 256 384 512; do
  GEMM_PACKED=1 python tools/gemm_bench.py $M 2>&1 | grep "gemm int8"
  XLLM_MI355_WS_ROWS128=0 GEMM_PACKED=1 python tools/gemm_bench.py $M 2>&1 | grep "gemm int8"
done | tee $O/gemm_rows.txt
