#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02diag2; mkdir -p $O; rm -f $O/*
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 --tb=line -k "dual_micro" 2>&1 | grep -v "Warning\|^$" | tail -4 >> $O/dual_alone.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 --tb=short -x 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -25 > $O/parity_file.txt
echo done > $O/done
