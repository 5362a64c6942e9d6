#!/bin/bash
# PMC-backed bound of the prefill W8A8 GEMM (gate_up at M = 8192): matrix-pipe busy cycles and the clock the chip grants, for three
# operand distributions (the power draw of the MACs depends on how many bits toggle)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/s26
O=$R/gpurun_out/s26
cd /tmp && export TMPDIR=/tmp
: > $O/pmc.txt
for D in uniform gauss zero; do
  for GU in 0 1; do
    echo "== GEMM_DIST=$D GEMM_GU=$GU  (python tools/gemm_one.py 8192 37888 3584, 8 launches)" >> $O/pmc.txt
    GEMM_LAUNCHES=8 GEMM_DIST=$D GEMM_GU=$GU rocprofv3 --kernel-trace -d $O/kt -- python $R/tools/gemm_one.py 8192 37888 3584 > /dev/null 2>&1
    python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) | grep "gemm_p8\|quantize_with" >> $O/pmc.txt
    rm -rf $O/kt
    GEMM_LAUNCHES=8 GEMM_DIST=$D GEMM_GU=$GU rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pm -- python $R/tools/gemm_one.py 8192 37888 3584 > /dev/null 2>&1
    python $R/tools/rocpd_summary.py $(find $O/pm -name "*.db" | head -1) --pmc | grep "gemm_p8" | grep "sum=" >> $O/pmc.txt
    rm -rf $O/pm
  done
done
cat $O/pmc.txt | cut -c1-60,100-260
