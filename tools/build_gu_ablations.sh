#!/bin/bash
# ablation builds of the gate_up epilogue of gemm_ws.hip (timing only, WRONG results): libxllm_mi355_gu_<name>.so (XLLM_MI355_LIB)
set -e
cd $(dirname $0)/../xllm_amd/csrc
for v in NOATOMIC NOLDSATOMIC; do
  mkdir -p build_gu_$v
  for f in build/*.o; do b=$(basename $f); [ $b = gemm_ws.o ] || cp $f build_gu_$v/$b; done
  D="-DWS_ABL_GU_$v"; [ $v = NOLDSATOMIC ] && D="$D -DWS_ABL_GU_NOATOMIC"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $D -c gemm_ws.hip -o build_gu_$v/gemm_ws.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libxllm_mi355_gu_$v.so build_gu_$v/*.o
  echo built gu_$v
done
