#!/usr/bin/env python3
"""A/B of the decode-attention launch at cfg3 inside ONE process: plain vs int8-epilogue kernel, 3 vs 28 rotating KV copies
(28 = one region per layer, 60 GB), variants interleaved over several rounds so that clock / thermal drift hits all alike."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
B, nq, nkv, S, d, bs = 256, 28, 4, 4096, 128, 128
pages = S // bs
nb = B * pages + 7
NC = int(os.environ.get("ATTN_COPIES", "28"))
caches = [(torch.randn(nb, bs, nkv, d, device=dev).bfloat16(), torch.randn(nb, bs, nkv, d, device=dev).bfloat16())
          for _ in range(NC)]
table = torch.randperm(nb, device=dev)[: B * pages].to(torch.int32).view(B, pages)
kv_lens = torch.full((B,), S, dtype=torch.int32, device=dev)
q = torch.randn(B, nq, d, device=dev).bfloat16()
plain = lambda c: ops.paged_attention(q, c[0], c[1], None, kv_lens, table, 1, S, d ** -0.5)
fused = lambda c: ops.paged_decode_attention_int8(q, c[0], c[1], kv_lens, table, S, d ** -0.5)


def unfused(c):
    o = ops.paged_attention(q, c[0], c[1], None, kv_lens, table, 1, S, d ** -0.5)
    return ops.scaled_quantize(o.view(B, nq * d))


variants = {"plain": plain, "int8_epilogue": fused, "plain+scaled_quantize": unfused}
bytes_ = B * (S * nkv * d * 4 + 4 * nq * d)
for copies in (3, NC):
    for name, fn in variants.items():
        for i in range(3):
            fn(caches[i % copies])
    for rnd in range(int(os.environ.get("ATTN_ROUNDS", "4"))):
        for name, fn in variants.items():
            n = 2 * NC
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn(caches[i % copies])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            print(f"[attn_ab] copies={copies:2d} round={rnd} {name:22s} {us:8.1f} us  {bytes_ / us / 1e3:7.1f} GB/s "
                  f"{bytes_ / us / 1e3 / 80:5.1f}%", flush=True)
