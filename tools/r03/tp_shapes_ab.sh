#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s33
O=gpurun_out/s33
B="python bench.py --no-cpu-baseline --no-engine --no-gemm --no-prefill --steps 20 --warmup 5"
run() { echo -n "$1 : "; env $2 $B $3 2>&1 | grep ms_per_step | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for k in 4 2; do
  run "tp$k default" "X=1" "--emulate-tp $k"
  run "tp$k rows128-all" "XLLM_MI355_WS_ROWS128=1" "--emulate-tp $k"
  run "tp$k unfused-gate_up" "XLLM_MI355_GATE_UP_FUSION=0" "--emulate-tp $k"
  run "tp$k unfused+rows128" "XLLM_MI355_GATE_UP_FUSION=0 XLLM_MI355_WS_ROWS128=1" "--emulate-tp $k"
done
done | tee $O/tp_ab.txt
