#!/bin/bash
# round-3 GPU session 4: PMC view of the packed decode GEMM at M = 256 (gate_up): where do the activation reads hit?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
O=$R/gpurun_out/s4
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TC[CP]_[A-Z0-9_]*" $O/counters.txt | sort -u | tr '\n' ' ' > $O/tc_names.txt
export GEMM_PACKED=1 XLLM_MI355_PACKED=1
L=$R/xllm_amd/lib
for lib in "" ws_NOWDMA ws_NOADMA; do
  if [ -n "$lib" ]; then export XLLM_MI355_LIB=$L/libxllm_mi355_$lib.so; else unset XLLM_MI355_LIB; fi
  for c in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum"; do
    n=$(echo "$lib $c" | tr ' ' '_' | cut -c1-48)
    timeout 200 rocprofv3 --pmc $c -d $O/pmc_$n -- python $R/tools/gemm_one.py 256 37888 3584 > $O/log_$n.txt 2>&1
    db=$(find $O/pmc_$n -name "*.db" | head -1)
    echo "== lib=[$lib] counters=[$c]"
    [ -n "$db" ] && python $R/tools/rocpd_summary.py $db --pmc 2>&1 | grep -E "gemm_ws" | grep -E "per_dispatch|gemm_ws_kernel" | awk '{print $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
    rm -rf $O/pmc_$n
  done
done > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
cat $O/tc_names.txt
