#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02last; mkdir -p $O; rm -f $O/*
timeout 1700 python -m pytest tests -q -m gpu --timeout 600 --tb=short 2>&1 | grep -v "Warning\|TypedDict\|^$" | tail -25 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo done > $O/done
