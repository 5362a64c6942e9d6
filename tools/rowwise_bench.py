#!/usr/bin/env python3
"""micro-benchmark of the row-wise operators around the GEMMs at the decode (M=256) and prefill (M=8192) shapes of
Qwen2-7B; inputs rotate over copies larger than L2 + Infinity Cache so that every launch reads HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
H, I, NQ, NKV, D = 3584, 18944, 28, 4, 128


def timeit(name, fn, nbytes, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"[rowwise] {name:34s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s")


for M in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["256", "8192"])]:
    copies = max(2, min(16, int(800e6 // (M * 2 * I * 2)) + 1))
    gu = [torch.randn(M, 2 * I, device=dev).bfloat16() for _ in range(copies)]
    timeit(f"M={M} silu_mul+int8quant", lambda i: ops.act_and_mul_dynamic_int8_quant(gu[i % copies], "silu"),
           M * (2 * I * 2 + I + 4))
    out = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    timeit(f"M={M} silu_mul", lambda i: ops.act_and_mul(out, gu[i % copies], "silu"), M * (2 * I * 2 + I * 2))
    del gu
    copies = max(2, min(64, int(800e6 // (M * H * 2)) + 1))
    xs = [torch.randn(M, H, device=dev).bfloat16() for _ in range(copies)]
    rs = [torch.randn(M, H, device=dev).bfloat16() for _ in range(copies)]
    w = torch.rand(H, device=dev).bfloat16()
    timeit(f"M={M} add+rmsnorm+int8quant", lambda i: ops.rms_norm_dynamic_int8_quant(xs[i % copies], w, 1e-6, residual=rs[i % copies]),
           M * (H * 2 * 3 + H + 4))
    timeit(f"M={M} rmsnorm+int8quant", lambda i: ops.rms_norm_dynamic_int8_quant(xs[i % copies], w, 1e-6), M * (H * 2 + H + 4))
    timeit(f"M={M} add+rmsnorm", lambda i: ops.fused_add_rms_norm(xs[i % copies], rs[i % copies], w, 1e-6), M * H * 2 * 4)
    timeit(f"M={M} int8quant", lambda i: ops.scaled_quantize(xs[i % copies]), M * (H * 2 + H + 4))
    qkv = [torch.randn(M, (NQ + 2 * NKV) * D, device=dev).bfloat16() for _ in range(copies)]
    pos = torch.randint(0, 4096, (M,), device=dev, dtype=torch.int64)
    cs = torch.randn(8192, D, device=dev).bfloat16()
    def rope(i):
        t = qkv[i % copies]
        ops.rotary_embedding(pos, t[:, :NQ * D], t[:, NQ * D:(NQ + NKV) * D], cs, True, head_size=D)
    timeit(f"M={M} rope", rope, M * (NQ + NKV) * D * 2 * 2)
    nblk = (M + 127) // 128 + 1
    kc = torch.zeros(nblk, 128, NKV, D, device=dev, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    slots = torch.arange(M, device=dev, dtype=torch.int32)
    def ropec(i):
        t = qkv[i % copies]
        ops.rotary_embedding_and_cache(pos, t[:, :NQ * D], t[:, NQ * D:(NQ + NKV) * D], t[:, (NQ + NKV) * D:], cs, slots,
                                       kc, vc, D, True)
    timeit(f"M={M} rope+kv_write", ropec, M * ((NQ + NKV) * D * 2 * 2 + 2 * NKV * D * 2 * 2))
