#!/usr/bin/env python3
"""a few dispatches of the MLA prefill kernel at 1 x 8192 tokens, 128 heads (for rocprofv3 --pmc / --kernel-trace)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev, bs, H, L = "cuda", 64, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 8192
pages = L // bs
cache = torch.randn(pages + 3, bs, 1, 576, device=dev).bfloat16()
table = torch.randperm(pages + 3, device=dev)[:pages].to(torch.int32).view(1, pages)
cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
kv = torch.tensor([L], dtype=torch.int32, device=dev)
q = torch.randn(L, H, 576, device=dev).bfloat16()
for _ in range(4):
    out = ops.mla_prefill(q, cache, cu, kv, table, 512, 192 ** -0.5, L, is_causal=True)
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
