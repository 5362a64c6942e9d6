#!/usr/bin/env python3
"""micro-benchmark of the W8A8 / bf16 GEMM entry points on the decode and prefill shapes of Qwen2-7B.
Rotates over several weight copies so every launch streams its weights from HBM (not L2 / Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
shapes = [("qkv", 4608, 3584), ("o", 3584, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944)]
if os.environ.get("GEMM_SHAPES") == "dsv3":      # DeepSeek-V3 MLA projections of one TP = 8 rank (bench_slices.py cfg4-slice)
    shapes = [("q_a", 1536, 7168), ("q_b", 16 * 192, 1536), ("kv_a", 576, 7168), ("o", 7168, 16 * 128)]
if os.environ.get("GEMM_SHAPES_ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["GEMM_SHAPES_ONLY"].split(",")]
Ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["256"])]
kind = sys.argv[2] if len(sys.argv) > 2 else "int8"
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("XLLM_MI355") or k in ("GEMM_DIST", "GEMM_PACKED", "GEMM_COPIES", "GEMM_FUSED", "GEMM_GU"))
for M in Ms:
    for name, N, K in shapes + ([("lm_head", 152064, 3584)] if kind == "bf16" else []):
        copies = int(os.environ.get('GEMM_COPIES', 0)) or max(2, min(8, int(600e6 // (N * K)) + 1))
        if kind == "int8":
            if os.environ.get("GEMM_DIST", "uniform") == "gauss":
                # what W8A8 quantisation of Gaussian tensors yields (per-row amax -> 127): sigma ~ 127 / 4, most bits quiet
                def q(t):
                    return torch.round(t / (t.abs().amax(-1, keepdim=True) / 127.0)).to(torch.int8)
                ws = [q(torch.randn(N, K, device=dev)) for _ in range(copies)]
                a = q(torch.randn(M, K, device=dev))
            elif os.environ.get("GEMM_DIST") == "zero":     # no operand toggling: the clock the chip grants without the MAC power
                ws = [torch.zeros(N, K, dtype=torch.int8, device=dev) for _ in range(copies)]
                a = torch.zeros(M, K, dtype=torch.int8, device=dev)
            else:
                ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(copies)]
                a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
            a_s = torch.rand(M, device=dev)
            w_s = torch.rand(N, device=dev)
            if os.environ.get("GEMM_FUSED", "0") == "1" and N == 3584:
                # the row-parallel projections as the decode step runs them: GEMM -> (slabs) -> residual add + RMSNorm + int8 quant
                wps = [ops.pack_weight_i8(x) for x in ws]
                res = torch.randn(M, N, device=dev).bfloat16()
                nw = torch.ones(N, device=dev).bfloat16()
                fn = lambda i: ops.scaled_matmul_add_rms_norm(a, ws[i % copies], a_s, w_s, res, nw, 1e-6, None, quantize=True,
                                                              b_packed=wps[i % copies])
            elif os.environ.get("GEMM_GU", "0") in ("1", "2") and name == "gate_up":
                # gate_up as the MLP runs it: GEMM -> SiLU * mul -> per-token int8 quant; 1 = fused into the GEMM epilogue + one
                # quantising pass, 2 = the GEMM and the fused row-wise operator
                wps = [ops.pack_weight_i8(x) for x in ws]
                a_s.mul_(0.01); w_s.mul_(0.01)
                if os.environ["GEMM_GU"] == "1":
                    fn = lambda i: ops.scaled_matmul_silu_mul_quant(a, ws[i % copies], a_s, w_s, torch.bfloat16, None, b_packed=wps[i % copies])
                else:
                    fn = lambda i: ops.act_and_mul_dynamic_int8_quant(
                        ops.scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies]), "silu")
            elif os.environ.get("GEMM_PACKED", "0") == "1":
                wps = [ops.pack_weight_i8(x) for x in ws]
                fn = lambda i: ops.scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies])
            else:
                fn = lambda i: ops.scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16)
            bytes_ = N * K + M * K + M * N * 2
        elif kind == "fp8":
            ws = [(torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn) for _ in range(copies)]
            a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e4m3fn)
            a_s = torch.full((1,), 0.03, device=dev)
            w_s = torch.full((1,), 0.02, device=dev)
            if os.environ.get("GEMM_PACKED", "0") == "1":
                wps = [ops.pack_weight_fp8(x) for x in ws]
                fn = lambda i: ops.fp8_scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies])
            else:
                fn = lambda i: ops.fp8_scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16)
            bytes_ = N * K + M * K + M * N * 2
        else:
            ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(copies)]
            a = torch.randn(M, K, device=dev).bfloat16()
            if os.environ.get("GEMM_PACKED", "0") == "1":
                wps = [ops.pack_weight_16(x) for x in ws]
                fn = lambda i: ops.matmul(a, ws[i % copies], b_packed=wps[i % copies])
            else:
                fn = lambda i: ops.matmul(a, ws[i % copies])
            bytes_ = (N * K + M * K + M * N) * 2
        for i in range(3):
            fn(i)
        # GEMM_N launches per graph, GEMM_REPLAYS timed replays after as many untimed ones: long enough (>= 50 ms) for the clock
        # governor to settle (round 3: 20-launch runs scattered by +-10 %)
        n = int(os.environ.get("GEMM_N", "100"))
        reps = int(os.environ.get("GEMM_REPLAYS", "10"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("GEMM_GRAPH", "1") == "1":   # n launches replayed from ONE HIP graph: no host launch cost in the figure
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                fn(0)   # (per-stream scratch buffers are created outside the capture)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for i in range(n):
                        fn(i)
                for _ in range(reps):
                    g.replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
            torch.cuda.synchronize()
            n *= reps
        else:
            e0.record()
            for i in range(n):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"[gemm {kind}] {tag:40s} M={M:5d} {name:8s} N={N:6d} K={K:6d}  {us:8.1f} us  {bytes_ / us / 1e3:7.1f} GB/s  "
              f"{2 * M * N * K / us / 1e6:7.1f} TOP/s")
        del ws
