#!/usr/bin/env python3
"""micro-benchmark of the MoE expert path at SURVEY 8 cfg5 (Qwen3-MoE: H=2048, 128 experts top-8, moe_I=768, T=8192 tokens):
fused top-k -> index build -> expand (gather) -> grouped GEMM w13 -> SiLU*mul -> grouped GEMM w2 -> combine."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
T, E, topk, H, I = int(os.environ.get("MOE_T", 8192)), 128, 8, 2048, 768
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(T, H, device=dev, generator=g).bfloat16()
logits = torch.randn(T, E, device=dev, generator=g).bfloat16()
w13 = (torch.randn(E, 2 * I, H, device=dev, generator=g) / 45).bfloat16()
w2 = (torch.randn(E, H, I, device=dev, generator=g) / 28).bfloat16()


def timed(fn, n=10):
    for _ in range(3):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


us_topk, (w, ids) = timed(lambda: ops.moe_fused_topk(logits, topk, True))
us_idx, (src_dst, dst_src, sizes) = timed(lambda: ops.moe_compute_index(ids, E))
us_exp, xs = timed(lambda: x.index_select(0, (dst_src // topk).long()))
us_g1, h13 = timed(lambda: ops.group_gemm(xs, w13, sizes))
us_g1g, h13g = timed(lambda: ops.group_gemm_gather(x, dst_src, topk, w13, sizes))
assert h13g is None or torch.equal(h13g, h13)
act = torch.empty(T * topk, I, dtype=torch.bfloat16, device=dev)
us_act, _ = timed(lambda: ops.act_and_mul(act, h13, "silu"))
us_g2, h2 = timed(lambda: ops.group_gemm(act, w2, sizes))
h2u = torch.empty_like(h2)
us_unsort, _ = timed(lambda: h2u.index_copy_(0, dst_src.long(), h2))
us_comb, out = timed(lambda: ops.moe_combine_result(h2u, w, T, topk))
us_fused, out2 = timed(lambda: ops.moe_combine_sorted(h2, src_dst, w, T, topk))
assert torch.equal(out, out2)
f1, f2 = 2 * T * topk * 2 * I * H, 2 * T * topk * H * I
g1g_txt = f"{us_g1g:.0f} us ({f1 / us_g1g / 1e6:.0f} TF/s)" if h13g is not None else "n/a (falls back)"
tot = us_topk + us_idx + (us_g1g if h13g is not None else us_exp + us_g1) + us_act + us_g2 + us_fused
print(f"[moe cfg5] T={T}: topk {us_topk:.0f} | index {us_idx:.0f} | (expand {us_exp:.0f} + w13 {us_g1:.0f} ->) w13 with gather {g1g_txt} | "
      f"act {us_act:.0f} | w2 {us_g2:.0f} us ({f2 / us_g2 / 1e6:.0f} TF/s) | (unsort {us_unsort:.0f} + combine {us_comb:.0f} ->) fused combine {us_fused:.0f} | "
      f"total {tot:.0f} us = {(f1 + f2) / tot / 1e6:.0f} TF/s   expert sizes min/max {int(sizes.min())}/{int(sizes.max())}")

# ---- W8A8 experts (config 5 names a grouped QUANT GEMM): per-token int8 activations, per-channel int8 expert weights
def q8(t):
    sc = t.abs().amax(-1, keepdim=True).clamp_min(1e-8) / 127.0
    return torch.round(t / sc).clamp_(-127, 127).to(torch.int8), sc.squeeze(-1).float()


w13q, w13s = q8(w13.float())
w2q, w2s = q8(w2.float())
us_q1, (xq, xs) = timed(lambda: ops.scaled_quantize(x))
us_i1, h13i = timed(lambda: ops.group_gemm_w8a8(xq, xs, w13q, w13s, sizes, row_index=dst_src, index_div=topk))
us_aq, (aq, as_) = timed(lambda: ops.act_and_mul_dynamic_int8_quant(h13i, "silu"))
us_i2, h2i = timed(lambda: ops.group_gemm_w8a8(aq, as_, w2q, w2s, sizes))
us_ci, outi = timed(lambda: ops.moe_combine_sorted(h2i, src_dst, w, T, topk))
toti = us_topk + us_idx + us_q1 + us_i1 + us_aq + us_i2 + us_ci
err = ((outi.float() - out.float()).norm() / out.float().norm()).item()
print(f"[moe cfg5 W8A8] T={T}: quant {us_q1:.0f} | w13 int8 with gather {us_i1:.0f} us ({f1 / us_i1 / 1e6:.0f} TOP/s) | silu*mul+quant {us_aq:.0f} | "
      f"w2 int8 {us_i2:.0f} us ({f2 / us_i2 / 1e6:.0f} TOP/s) | combine {us_ci:.0f} | total {toti:.0f} us = {(f1 + f2) / toti / 1e6:.0f} TOP/s   "
      f"(vs the 16-bit path: relative L2 difference {err:.1e})")
