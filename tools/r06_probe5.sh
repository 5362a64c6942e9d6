#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_probe5.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "moe or group" -p no:cacheprovider 2>&1 | tail -5 >> $O
timeout 600 python tools/moe_bench.py 2>&1 | grep "^\[moe" >> $O
timeout 600 python bench.py --config cfg5-slice --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_bench_cfg5_slice.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_cfg5_slice.json')); print(d['ms_per_step'], d['roofline'], d.get('w2_gemm'))" >> $O
cat $O
