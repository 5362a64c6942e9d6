export XLLM_MI355_P8=1
for v in "" _nostage _nomfma _noepi _nostage_noepi _only_sync; do
  if [ -n "$v" ]; then export XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355$v.so; fi
  echo "== variant [$v]"
  timeout 200 python tools/gemm_bench.py 8192 int8 2>&1 | grep gemm | awk '{print $(NF-10), $(NF-5), "us", $(NF-1), "TOP/s"}'
done
unset XLLM_MI355_LIB
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/p8pmc_$n -- python $GRAFT_REPO_ROOT/tools/gemm_one.py 8192 37888 3584 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/p8pmc_$n -name "*.db" | head -1) --pmc 2>&1 | grep -E "gemm_p8" | grep per_dispatch | awk '{print $(NF-4), $NF}'
done
