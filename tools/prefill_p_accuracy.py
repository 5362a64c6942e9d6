#!/usr/bin/env python3
"""accuracy of the flash prefill kernel's P operand mode (XLLM_MI355_PREFILL_P = 1: one 16-bit P, 2: hi + lo) against the
oracle with fp32 P and with P rounded to the tensor dtype before PV (p_round, layers/cuda/flashinfer_attention.cpp:84-90).
Test infrastructure: the oracle is the checker here."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from xllm_amd import ops  # noqa: E402

NQ, NKV, D = 28, 4, 128
lens = [700, 1024, 129]
T = sum(lens)
g = torch.Generator().manual_seed(3)
for sigma in (1.0, 3.0):
    qkv = (torch.randn(T, (NQ + 2 * NKV) * D, generator=g) * sigma).bfloat16()
    q = qkv[:, :NQ * D].unflatten(-1, (NQ, D))
    k = qkv[:, NQ * D:(NQ + NKV) * D].unflatten(-1, (NKV, D))
    v = qkv[:, (NQ + NKV) * D:].unflatten(-1, (NKV, D))
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    scale = 1.0 / math.sqrt(D)
    ref0 = orc.attention_varlen(q, k, v, cu, cu, scale, True, -1, False).float()
    ref1 = orc.attention_varlen(q, k, v, cu, cu, scale, True, -1, True).float()
    qd = qkv.cuda()
    out = ops.prefill_attention(qd[:, :NQ * D].unflatten(-1, (NQ, D)), qd[:, NQ * D:(NQ + NKV) * D].unflatten(-1, (NKV, D)),
                                qd[:, (NQ + NKV) * D:].unflatten(-1, (NKV, D)), cu.cuda(), cu.cuda(), max(lens), scale,
                                True).float().cpu().view(T, -1)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"[prefill P mode {os.environ.get('XLLM_MI355_PREFILL_P', 'default')}] sigma={sigma}: vs fp32-P oracle {rel(out, ref0):.3e}  "
          f"vs p_round oracle {rel(out, ref1):.3e}   (p_round oracle vs fp32-P oracle {rel(ref1, ref0):.3e})")
