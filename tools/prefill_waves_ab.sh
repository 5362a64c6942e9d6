T=$GRAFT_REPO_ROOT/xllm_amd/lib/libxllm_mi355_tuning.so
for rep in 1 2; do
for w in 4 8; do echo "waves=$w: $(XLLM_MI355_LIB=$T XLLM_MI355_PREFILL_M32_WAVES=$w python tools/prefill_attn_one.py 2>&1 | tail -1)"; done
done
echo "waves=8 P=2: $(XLLM_MI355_LIB=$T XLLM_MI355_PREFILL_M32_WAVES=8 XLLM_MI355_PREFILL_P=2 python tools/prefill_attn_one.py 2>&1 | tail -1)"
python -m pytest tests/test_gpu_parity.py -x -q -k "prefill or chunked" -p no:cacheprovider 2>&1 | tail -3
