#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s27
O=gpurun_out/s27
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -k "qkv_gemm_rope or add_rms_norm or prefill_parity_under or fused or golden or model" -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
python tools/step_ab.py 256 4096 base= rows128_all=ws_waves=128 rows128_small=ws_waves=129 rows128_qkvo=ws_waves=130 2>&1 | grep step_ab | tee $O/ab256.txt
python tools/step_ab.py 32 4096 base= 2>&1 | grep step_ab | tee $O/ab32.txt
