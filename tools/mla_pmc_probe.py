"""The MLA decode kernel alone, for a rocprofv3 --pmc pass launched BY bench.py --config cfg4-slice (roofline.traffic measured in the
same run): one TP-8 rank of DeepSeek-V3 (16 heads, latent 576, values 512, pages of 64 tokens), B sequences at ctx tokens over random
pages. Usage: mla_pmc_probe.py [B ctx heads launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

B, ctx, heads, n = (int(x) for x in (sys.argv[1:5] + ["128", "8192", "16", "4"][len(sys.argv) - 1:]))
kv_lora, rope, bs = 512, 64, 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
pages = ctx // bs
nb = B * pages + 8
cache = torch.empty(nb, bs, 1, kv_lora + rope, dtype=torch.bfloat16, device=dev).normal_(generator=g)
q = torch.empty(B, heads, kv_lora + rope, dtype=torch.bfloat16, device=dev).normal_(generator=g)
table = torch.randperm(nb, device=dev, generator=g)[:B * pages].to(torch.int32).view(B, pages)
lens = torch.full((B,), ctx, dtype=torch.int32, device=dev)
for _ in range(n):
    ops.mla_decode(q, cache, lens, table, kv_lora, (192.0) ** -0.5, ctx)
torch.cuda.synchronize()
print("probe done")
