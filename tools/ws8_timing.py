#!/usr/bin/env python3
"""per-phase shader cycles of gemm_ws8s_kernel (timing build: tools/build_ws8_timing.sh, XLLM_MI355_LIB=...ws8t.so)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("XLLM_MI355_PACKED", "1")
from xllm_amd import _lib, ops  # noqa: E402

M, N, K = [int(x) for x in sys.argv[1:4]]
w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
a_s = torch.rand(M, device="cuda")
w_s = torch.rand(N, device="cuda")
wp = ops.pack_weight_i8(w)
for _ in range(3):
    ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b_packed=wp)
torch.cuda.synchronize()
# a long run (default 4000 launches ~ 0.2 s) so that the clock governor reaches its steady state; wall time per launch
n = int(os.environ.get("WS8_N", "4000"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(100):
            ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b_packed=wp)
    for _ in range(n // 200):
        g.replay()
    e0.record()
    for _ in range(n // 200):
        g.replay()
    e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (n // 200 * 100)
print(f"[ws8t] M={M} N={N} K={K}: {us:.1f} us per launch in a {n}-launch run")
out = (C.c_longlong * 64)()
assert _lib.lib().xllm_mi355_debug_ws8(out) == 0
names = ["read+lgkmcnt(0)", "barrier A", "mfma+dma issue", "vmcnt wait", "barrier B"]
for sel, blk in ((0, "first"), (1, "middle")):
    for g in (0, 1):
        o = out[sel * 32 + g * 16: sel * 32 + g * 16 + 16]
        nk = max(o[5], 1)
        ghz = o[10] / max(o[9], 1) * 0.1
        print(f"[ws8t] M={M} N={N} K={K} {blk} block, group {g}: clock {ghz:.2f} GHz, whole workgroup {o[10]} cycles = {o[10] / ghz / 1e3:.1f} us "
              f"(prologue {o[6]}, K loop {o[7]}, epilogue {o[8]}); K tiles {nk}, cycles per tile {o[7] / nk:.0f}: " +
              ", ".join(f"{n} {x / nk:.0f}" for n, x in zip(names, o[:5])))
