#!/bin/bash
# builds ablation variants of libxllm_mi355.so (timing only, results are WRONG): tools/build_ablations.sh NAME "-DFLAG ..."
set -e
cd $(dirname $0)/../xllm_amd/csrc
name=$1; flags=$2
mkdir -p build_$name ../lib
for f in rowwise gemm gemm_p8 gemm_p8i gemm_ws gemm_wsb attention_decode attention_prefill attention_mla attention_api moe sampling logits_processors allreduce host_batch workspace build_info; do
  if [ $f = gemm_p8 ] || [ $f = gemm ] || [ $f = gemm_p8i ] || [ $f = attention_decode ] || [ $f = attention_prefill ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags -c $f.hip -o build_$name/$f.o
  else
    cp build/$f.o build_$name/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libxllm_mi355_$name.so build_$name/*.o
echo built ../lib/libxllm_mi355_$name.so
