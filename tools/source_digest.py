#!/usr/bin/env python3
"""One digest over every source the GPU suite exercises (kernels, C ABI, host mirror, shim, oracle, tests): the committed record
of the last full `pytest -m gpu` run carries it, and tests/test_profiles_records.py compares it with the tree (round-3 review:
the submitted HEAD must be the tree the full GPU suite ran on). Documentation, profiles and tools are not part of it."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ["xllm_amd/csrc/*.hip", "xllm_amd/csrc/*.h", "xllm_amd/csrc/Makefile", "xllm_amd/*.py", "include/*.h", "shim/*.cpp", "shim/*.h",
            "shim/build_shim.py", "shim/stub/**/*.h", "shim/stub/**/*.cpp", "oracle/*.py", "oracle/*.c", "oracle/Makefile", "tests/*.py", "tests/golden/*",
            "__graft_entry__.py"]


def files():
    out = set()
    for pat in PATTERNS:
        out.update(p for p in glob.glob(os.path.join(ROOT, pat), recursive=True) if os.path.isfile(p))
    return sorted(out)


def digest() -> str:
    h = hashlib.sha256()
    for p in files():
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def lib_digest() -> str:
    """what xllm_mi355_build_digest() must report for this tree (the recipe of xllm_amd/csrc/Makefile: DIGEST_SRCS)"""
    csrc = os.path.join(ROOT, "xllm_amd", "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for p in [os.path.join(csrc, n) for n in names] + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "xllm_mi355.h")]:
        h.update(open(p, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    import sys
    print(lib_digest() if "--lib" in sys.argv else digest())
