#!/usr/bin/env python3
"""micro-benchmark of the MLA decode kernel at SURVEY 8 cfg4 (DeepSeek-V3 per-rank shape: B=128, S=8192, 16 heads/rank,
latent cache [n_blocks, 64, 1, 576] bf16) and the TP=1 head count; rotates over 3 caches so every launch reads HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev, bs = "cuda", 64
for name, B, H, S in [("cfg4 (16 heads/rank)", 128, 16, 8192), ("128 heads", 128, 128, 8192), ("16 heads, B=32", 32, 16, 8192)]:
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
