#!/usr/bin/env python3
"""micro-benchmark of the fp8 (e4m3) W8A8 GEMM at the Qwen2-7B layer shapes; weights from a Gaussian tensor quantised per
tensor, activations per token. usage: fp8_gemm_bench.py [M,M,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
shapes = [("qkv", 4608, 3584), ("o", 3584, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944)]
Ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "256"])]
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("XLLM_MI355"))
for M in Ms:
    for name, N, K in shapes:
        copies = max(2, min(8, int(600e6 // (N * K)) + 1))
        ws = []
        for _ in range(copies):
            w = torch.randn(N, K, device=dev)
            ws.append((w / (w.abs().max() / 448.0)).to(torch.float8_e4m3fn))
        x = torch.randn(M, K, device=dev).bfloat16()
        a, a_s = ops.fp8_scaled_quantize(x)
        w_s = torch.tensor([0.01], device=dev)
        fn = lambda i: ops.fp8_scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16)
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"[gemm fp8] {tag:30s} M={M:5d} {name:8s} N={N:6d} K={K:6d} {us:9.1f} us  {(N * K + M * K + 2 * M * N) / us / 1e3:8.1f} GB/s  "
              f"{2 * M * N * K / us / 1e6:8.1f} TFLOP/s")
        del ws
