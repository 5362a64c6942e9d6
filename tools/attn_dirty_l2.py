#!/usr/bin/env python3
"""Does the state the preceding kernels leave in the L2 explain the ~9 us the decode-attention launch loses inside the step
? One process, per-launch HIP events around the attention kernel alone (cfg3 shape, 6 rotating KV copies), with a
'predecessor' launched right before each timed launch:
   none          nothing (the stand-alone loop of tools/attn_ab.py)
   dirty NN MB   a kernel that WRITES NN MB (the qkv GEMM's K-slice slabs + q / k / v rows leave ~17 MB of dirty lines in 32 MB of L2)
   read NN MB    a kernel that READS NN MB (clean lines only)
   gemm          the step's own predecessors: packed qkv GEMM (M = 256) + its fused RoPE / KV-write consumer
The predecessor is outside the timed events (event -> attention -> event); what is compared is the attention launch's duration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
B, nq, nkv, S, d, bs = 256, 28, 4, 4096, 128, 128
pages = S // bs
nb = B * pages + 7
NC = 6
caches = [(torch.randn(nb, bs, nkv, d, device=dev).bfloat16(), torch.randn(nb, bs, nkv, d, device=dev).bfloat16()) for _ in range(NC)]
table = torch.randperm(nb, device=dev)[: B * pages].to(torch.int32).view(B, pages)
kv_lens = torch.full((B,), S, dtype=torch.int32, device=dev)
q = torch.randn(B, nq, d, device=dev).bfloat16()
attn = lambda c: ops.paged_decode_attention_int8(q, c[0], c[1], kv_lens, table, S, d ** -0.5)
junk = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
# the step's predecessors
H = 3584
xq = torch.randint(-127, 128, (B, H), dtype=torch.int8, device=dev)
xs = torch.rand(B, device=dev) * 0.01 + 0.001
w = torch.randint(-127, 128, ((nq + 2 * nkv) * d, H), dtype=torch.int8, device=dev)
wp = ops.pack_weight_i8(w)
w_s = torch.rand((nq + 2 * nkv) * d, device=dev) * 0.01 + 0.001
pos = torch.full((B,), S - 1, dtype=torch.int64, device=dev)
cos_sin = torch.randn(8192, d, device=dev).bfloat16()
slots = (table[:, -1].long() * bs + bs - 1).to(torch.int32)


def pred_gemm(c):
    ops.scaled_matmul_rope_cache(xq, wp, xs, w_s, None, pos, cos_sin, slots, c[0], c[1], nq, nkv, d)


preds = {"none": lambda c: None,
         "dirty 17 MB": lambda c: junk[:17 << 20].fill_(1), "dirty 32 MB": lambda c: junk[:32 << 20].fill_(1),
         "read 32 MB": lambda c: junk[:32 << 20].sum(), "gemm+rope (the step's own)": pred_gemm}
for name, pred in preds.items():
    for i in range(3):
        pred(caches[i % NC]); attn(caches[i % NC])
torch.cuda.synchronize()
res = {k: [] for k in preds}
for rnd in range(5):
    for name, pred in preds.items():
        ev = []
        for i in range(2 * NC):
            c = caches[i % NC]
            pred(c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); attn(c); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        res[name].append(t[len(t) // 2])
for name, v in res.items():
    print(f"[attn_dirty_l2] predecessor {name:28s}: attention {sum(v) / len(v):7.1f} us (rounds: {' '.join('%.1f' % x for x in v)})")
