#!/bin/bash
# the round's LAST GPU run: the full suite on exactly the tree that is submitted. The log starts with the digest of the sources
# it ran on (tools/source_digest.py); profiles/rNN_pytest_gpu.txt (the latest one) is a copy of it and tests/test_profiles_records.py holds the
# committed tree to that digest.
# (Serial on purpose: `pytest -n 3` (pytest-xdist) was tried at the end of round 4 and is SLOWER -- 201 tests in 420 s against 430 in
# 527 s serial: the oracle's OpenMP teams and the full-size tests' host work contend for the box's cores.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/final
O=gpurun_out/final/pytest_gpu.txt
echo "# source-digest: $(python tools/source_digest.py)" > $O
echo "# python -m pytest tests -m gpu -x -q   ($(date -u +%Y-%m-%dT%H:%M:%SZ), $(rocminfo 2>/dev/null | grep -m1 gfx9 | xargs))" >> $O
python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=30 >> $O 2>&1
echo "# pytest rc=$?" >> $O
tail -4 $O
