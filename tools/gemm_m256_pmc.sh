#!/bin/bash
# PMC evidence for the decode-shaped gate_up GEMM (M = 256, packed weights, SiLU.mul epilogue): matrix-pipe busy + granted clock in one
# pass, HBM traffic (FETCH_SIZE / WRITE_SIZE) in two more (one counter per pass, as MI355X_MICROARCH.md prescribes).
# Output: gpurun_out/gemm_m256_pmc.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gemm_m256_pmc.txt
: > $O
cd /tmp && export TMPDIR=/tmp
export GEMM_PACKED=1 GEMM_GU=1 GEMM_DIST=gauss GEMM_LAUNCHES=8 XLLM_MI355_PACKED=1
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pmc_out -- python $R/tools/gemm_one.py 256 37888 3584 > /dev/null 2>&1
  echo "## --pmc $pmc   (tools/gemm_one.py 256 37888 3584, GEMM_PACKED=1 GEMM_GU=1 GEMM_DIST=gauss)" >> $O
  python $R/tools/rocpd_summary.py $(find /tmp/pmc_out -name "*.db" | head -1) --pmc 2>&1 | grep -i "gemm_ws8s\|quantize_with\|^kernel" | cut -c1-260 >> $O
done
python - <<PY >> $O
import re
t = open("$O").read()
def per(name, kern="gemm_ws8s"):
    m = re.search(kern + r".*?" + name + r"\s+sum=\S+ dispatches=\d+ per_dispatch=(\S+)", t)
    return float(m.group(1)) if m else None
busy, act = per("SQ_VALU_MFMA_BUSY_CYCLES"), per("GRBM_GUI_ACTIVE")
dur = re.search(r"gemm_ws8s_kernel\S*\s+(\d+)\s+\S+\s+(\S+)", t)
f, w = per("FETCH_SIZE"), per("WRITE_SIZE")
print("## derived")
if busy and act:
    cyc = act / 8.0
    print(f"matrix-pipe busy = {busy / (cyc * 1024):.3f} (MFMA-busy cycles / (active cycles per XCD x 1024 SIMDs)); active cycles per XCD {cyc:.0f}")
    if dur:
        us = float(dur.group(2))
        print(f"kernel avg {us:.1f} us under PMC -> granted clock {cyc / us / 1e3:.2f} GHz")
if f and w:
    hbm = 2 * f * 1024 + w * 1024
    alg = 37888 * 3584 + 256 * 3584 + 256 * 18944 * 2
    print(f"HBM traffic per launch = {hbm / 1e6:.1f} MB (FETCH_SIZE x 2 + WRITE_SIZE) = {hbm / alg:.3f} x algorithmic ({alg / 1e6:.1f} MB: weights + activations + act out)")
PY
cat $O
