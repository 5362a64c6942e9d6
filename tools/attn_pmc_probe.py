"""The dominant kernel alone, for a rocprofv3 --pmc pass launched BY bench.py (roofline.traffic measured in the same run):
paged decode attention at the bench's shape (cfg3: B = 256, ctx = 4096, nq 28 / nkv 4 / d 128, pages of 128 tokens, fused int8
epilogue as in the step), a handful of eager launches over random pages. Usage: attn_pmc_probe.py [B ctx nq nkv launches]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

B, ctx, nq, nkv, n = (int(x) for x in (sys.argv[1:6] + ["256", "4096", "28", "4", "4"][len(sys.argv) - 1:]))
d, bs = 128, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
pages = ctx // bs
nb = B * pages + 8
kc = torch.empty(nb, bs, nkv, d, dtype=torch.bfloat16, device=dev).normal_(generator=g)
vc = torch.empty(nb, bs, nkv, d, dtype=torch.bfloat16, device=dev).normal_(generator=g)
q = torch.empty(B, nq, d, dtype=torch.bfloat16, device=dev).normal_(generator=g)
table = torch.randperm(nb, device=dev, generator=g)[:B * pages].to(torch.int32).view(B, pages)
lens = torch.full((B,), ctx, dtype=torch.int32, device=dev)
for _ in range(n):
    out = ops.paged_decode_attention_int8(q, kc, vc, lens, table, ctx, 1.0 / math.sqrt(d))
    if out is None:
        ops.paged_attention(q, kc, vc, None, lens, table, 1, ctx, 1.0 / math.sqrt(d))
torch.cuda.synchronize()
print("probe done")
