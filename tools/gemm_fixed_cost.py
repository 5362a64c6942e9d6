#!/usr/bin/env python3
"""Fixed cost vs K-loop cost of the decode-shaped packed GEMMs (product library, no special build): the same launch at several K
(K / 128 tiles per workgroup, same grid, same epilogue) timed in a graph over rotating weight copies; a straight-line fit gives
us_per_K_tile and the K-independent remainder (launch + ramp + prologue + epilogue + tail). Usage: gemm_fixed_cost.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


def bench(fn, n=50, reps=8):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(0)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(n):
                fn(i)
        for _ in range(reps):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


for name, N, fused in (("gate_up+SiLU.mul (+quant pass)", 37888, True), ("gate_up plain", 37888, False), ("qkv-like N=4608 plain", 4608, False)):
    pts = []
    for K in (512, 1024, 2048, 3584, 7168):
        copies = max(2, min(6, (600 << 20) // (N * K)))
        ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g) for _ in range(copies)]
        wps = [ops.pack_weight_i8(w) for w in ws]
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        a_s = torch.rand(M, device=dev, generator=g) * 0.01
        w_s = torch.rand(N, device=dev, generator=g) * 0.01
        if fused:
            fn = lambda i: ops.scaled_matmul_silu_mul_quant(a, ws[i % copies], a_s, w_s, torch.bfloat16, None, b_packed=wps[i % copies])
        else:
            fn = lambda i: ops.scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies])
        if fn(0) is None:
            continue
        pts.append((K // 128, bench(fn)))
        del ws, wps
        torch.cuda.empty_cache()
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] ** 2 for p in pts), sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
    print(f"[fixed] M={M} {name}: " + "  ".join(f"K={k * 128}: {us:.1f} us" for k, us in pts) +
          f"  => {slope:.3f} us per K tile, {icpt:.1f} us K-independent (at K = 3584: {28 * slope:.1f} us of K loop)")
