#!/usr/bin/env python3
"""micro-benchmark of the MLA decode kernel at SURVEY 8 cfg4 (DeepSeek-V3 per-rank shape: B=128, S=8192, 16 heads/rank,
latent cache [n_blocks, 64, 1, 576] bf16) and the TP=1 head count; rotates over 3 caches so every launch reads HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev, bs = "cuda", 64
DECODE = [] if "prefill-only" in sys.argv[1:] else [("cfg4 (16 heads/rank)", 128, 16, 8192), ("128 heads", 128, 128, 8192), ("16 heads, B=32", 32, 16, 8192)]
for name, B, H, S in DECODE:
    pages = S // bs
    nb = B * pages + 5
    caches = [torch.randn(nb, bs, 1, 576, device=dev).bfloat16() for _ in range(3)]
    table = torch.randperm(nb, device=dev)[: B * pages].to(torch.int32).view(B, pages)
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    q = torch.randn(B, H, 576, device=dev).bfloat16()
    fn = lambda i: ops.mla_decode(q, caches[i % 3], lens, table, 512, 192 ** -0.5, S)
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    by = B * S * 576 * 2
    fl = 2 * B * H * S * (576 + 512)
    print(f"[mla decode] {name:22s} B={B} H={H} S={S}: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s ({by / us / 1e3 / 80:4.1f}% of 8 TB/s)  "
          f"{fl / us / 1e6:6.1f} TFLOP/s")
    del caches


def prefill_leg():
    """MLA prefill / chunked prefill (ops.mla_prefill); the kernel choice is read once per process:
    XLLM_MI355_MLA_PREFILL=0 -> one decode-kernel entry per query token, default -> four query tokens share a KV tile"""
    print(f"[mla prefill] XLLM_MI355_MLA_PREFILL={os.environ.get('XLLM_MI355_MLA_PREFILL', '(default)')}")
    for name, H, lens in [("1 x 8192, 128 heads", 128, [(8192, 8192)]), ("1 x 8192, 16 heads", 16, [(8192, 8192)]),
                          ("4 x 2048, 128 heads", 128, [(2048, 2048)] * 4), ("chunk 1024 of 8192, 128 heads", 128, [(1024, 8192)]),
                          ("8 x (256 of 4096), 16 heads", 16, [(256, 4096)] * 8)]:
        B = len(lens)
        pages = [(kv + bs - 1) // bs for _, kv in lens]
        nb = sum(pages) + 5
        cache = torch.randn(nb, bs, 1, 576, device=dev).bfloat16()
        perm = torch.randperm(nb, device=dev).to(torch.int32)
        table = torch.zeros(B, max(pages), dtype=torch.int32, device=dev)
        used = 0
        for b, n in enumerate(pages):
            table[b, :n] = perm[used:used + n]
            used += n
        T = sum(ql for ql, _ in lens)
        cu = torch.tensor([0] + list(torch.tensor([ql for ql, _ in lens]).cumsum(0)), dtype=torch.int32, device=dev)
        kv = torch.tensor([k for _, k in lens], dtype=torch.int32, device=dev)
        q = torch.randn(T, H, 576, device=dev).bfloat16()
        fn = lambda: ops.mla_prefill(q, cache, cu, kv, table, 512, 192 ** -0.5, max(k for _, k in lens), is_causal=True)
        reps = 3 if os.environ.get("XLLM_MI355_MLA_PREFILL") == "0" else 10
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        keys = sum(ql * (k - ql) + ql * (ql + 1) // 2 for ql, k in lens)    # attended (query, key) pairs per head
        fl = 2 * H * keys * (576 + 512)
        print(f"[mla prefill] {name:32s} T={T:5d}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")


if "prefill" in sys.argv[1:] or "prefill-only" in sys.argv[1:]:
    prefill_leg()
