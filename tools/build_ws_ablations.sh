#!/bin/bash
# ablation builds of gemm_ws.hip (timing only, WRONG results): libxllm_mi355_<name>.so, selected with XLLM_MI355_LIB
set -e
cd $(dirname $0)/../xllm_amd/csrc
for v in NOADMA NOWDMA NOCOMPUTE NOMFMA; do
  mkdir -p build_ws_$v
  for f in build/*.o; do b=$(basename $f); [ $b = gemm_ws.o ] || cp $f build_ws_$v/$b; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DWS_ABL_$v -c gemm_ws.hip -o build_ws_$v/gemm_ws.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libxllm_mi355_ws_$v.so build_ws_$v/*.o
  echo built ws_$v
done
