#!/bin/bash
# timing build of attention_prefill.hip (per-phase shader cycles of one wave of flash_prefill_m32_kernel): libxllm_mi355_pf32time.so,
# loaded through XLLM_MI355_LIB by tools/pf32_timing.py; PF32_DEFS=-DPF32_TRACE PF32_TAG=trace builds the per-workgroup trace flavour (tools/pf32_trace.py)
set -e
cd $(dirname $0)/../xllm_amd/csrc
make -s tuning
mkdir -p build_pf32t
for f in build_tuning/*.o; do b=$(basename $f); [ $b = attention_prefill.o ] || cp $f build_pf32t/$b; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DXM_TUNING ${PF32_DEFS:--DPF32_TIMING} $PF32_EXTRA -c attention_prefill.hip -o build_pf32t/attention_prefill.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libxllm_mi355_pf32${PF32_TAG:-time}.so build_pf32t/*.o
echo built pf32${PF32_TAG:-time}
