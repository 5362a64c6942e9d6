#!/bin/bash
# timing build of gemm_ws.hip (per-phase shader cycles of the staggered eight-wave tile): libxllm_mi355_ws8t.so, XLLM_MI355_LIB
set -e
cd $(dirname $0)/../xllm_amd/csrc
mkdir -p build_ws8t
for f in build/*.o; do b=$(basename $f); [ $b = gemm_ws.o ] || cp $f build_ws8t/$b; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DWS8_TIMING $WS8_EXTRA -c gemm_ws.hip -o build_ws8t/gemm_ws.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libxllm_mi355_ws8t${WS8_TAG}.so build_ws8t/*.o
echo built ws8t${WS8_TAG}
