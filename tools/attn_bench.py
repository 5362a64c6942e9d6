#!/usr/bin/env python3
"""micro-benchmark of the paged decode attention kernel on the per-rank shapes of Qwen2-7B (TP = 1, 2, 4) and cfg2.
Rotates over 3 KV-cache copies (each >> Infinity Cache) so every launch streams its KV from HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
cases = [("tp1", 256, 28, 4, 4096), ("tp2", 256, 14, 2, 4096), ("tp4", 256, 7, 1, 4096), ("cfg2", 64, 28, 4, 2048),
         ("tp4dp2", 128, 7, 1, 4096), ("dp2", 128, 28, 4, 4096), ("dp4", 64, 28, 4, 4096), ("dp8", 32, 28, 4, 4096)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1].split(",")]
tag = " ".join(f"{k[11:]}={v}" for k, v in os.environ.items() if k.startswith("XLLM_MI355"))
d, bs = 128, 128
for name, B, nq, nkv, S in cases:
    pages = S // bs
    nb = B * pages + 7
    mk = (lambda: torch.zeros(nb, bs, nkv, d, device=dev, dtype=torch.bfloat16)) if os.environ.get("ATTN_KV") == "zero" \
        else (lambda: torch.randn(nb, bs, nkv, d, device=dev).bfloat16())
    NC = int(os.environ.get("ATTN_COPIES", "3"))   # 28 = as many distinct KV regions as the model has layers (60 GB at cfg3)
    caches = [(mk(), mk()) for _ in range(NC)]
    if os.environ.get("ATTN_PAGES") == "linear":   # pages in address order: is the random page placement a cost?
        table = torch.arange(B * pages, device=dev, dtype=torch.int32).view(B, pages)
    else:
        table = torch.randperm(nb, device=dev)[: B * pages].to(torch.int32).view(B, pages)
    kv_lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    q = torch.randn(B, nq, d, device=dev).bfloat16()
    if os.environ.get("ATTN_INT8") == "1":          # the bench path: epilogue emits o_proj's int8 operand
        def fn(i):   # what the layer runs: the fused epilogue, or -- when the plan declines it -- attention + scaled_quantize
            got = ops.paged_decode_attention_int8(q, caches[i % NC][0], caches[i % NC][1], kv_lens, table, S, d ** -0.5)
            if got is None:
                got = ops.scaled_quantize(ops.paged_attention(q, caches[i % NC][0], caches[i % NC][1], None, kv_lens, table, 1, S, d ** -0.5))
            return got
    else:
        fn = lambda i: ops.paged_attention(q, caches[i % NC][0], caches[i % NC][1], None, kv_lens, table, 1, S, d ** -0.5)
    for i in range(3):
        fn(i)
    n = int(os.environ.get("ATTN_N", "20"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    bytes_ = B * (S * nkv * d * 4 + 4 * nq * d)
    print(f"[attn] {tag:24s} {name:7s} B={B} nq={nq} nkv={nkv} S={S}: {us:8.1f} us  {bytes_ / us / 1e3:7.1f} GB/s  "
          f"{bytes_ / us / 1e3 / 80:5.1f}% of 8 TB/s")
    del caches
