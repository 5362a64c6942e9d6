#!/usr/bin/env python3
"""Generates tests/golden/operators_v1.npz: small seeded inputs + the CPU oracle's outputs for every operator of the
hot path. The fixtures pin the oracle against drift (tests/test_golden_fixtures.py, CPU) and are what the HIP kernels
are compared with on the GPU box, where /root/reference does not exist.  16-bit tensors are stored as their raw bits
(int16) with a `<name>__dtype` tag.  Regenerate only on purpose:  python tests/golden/make_golden.py
The reference's OWN golden vectors (hard-coded in its tests) live next to it in reference_goldens.json."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "operators_v1.npz")


def cases():
    g = torch.Generator().manual_seed(20260923)
    c = {}
    bf = torch.bfloat16
    # --- fixture generator of the reference tests (tests_utils.cpp:159-274)
    c["seeded_tensor.out"] = orc.seeded_tensor("golden.seeded", (4, 5), torch.float32)
    # --- RMSNorm family
    x = (torch.randn(5, 256, generator=g) * 2).to(bf)
    w = (torch.rand(256, generator=g) + 0.5).to(bf)
    res = torch.randn(5, 256, generator=g).to(bf)
    o = torch.empty_like(x)
    orc.rms_norm(o, x, w, 1e-6)
    c.update({"rms_norm.x": x, "rms_norm.w": w, "rms_norm.out": o})
    x2, r2 = x.clone(), res.clone()
    orc.fused_add_rms_norm(x2, r2, w, 1e-6)
    c.update({"fused_add_rms_norm.res_in": res, "fused_add_rms_norm.out": x2, "fused_add_rms_norm.res_out": r2})
    # --- RoPE (neox), positions int64
    nq, nk, d = 4, 2, 64
    q = torch.randn(6, nq * d, generator=g).to(bf)
    k = torch.randn(6, nk * d, generator=g).to(bf)
    pos = torch.tensor([0, 1, 5, 17, 100, 2047])
    cache = orc.build_cos_sin_cache(2048, d, 1000000.0, bf)
    q2, k2 = q.clone(), k.clone()
    orc.rotary_embedding(pos, q2, k2, cache, d, is_neox=True)
    c.update({"rope.q": q, "rope.k": k, "rope.pos": pos, "rope.cache": cache, "rope.q_out": q2, "rope.k_out": k2})
    # --- SiLU * mul, int8 per-token quant
    gu = (torch.randn(4, 256, generator=g) * 2).to(bf)
    act = torch.empty(4, 128, dtype=bf)
    orc.act_and_mul(act, gu, "silu")
    qa, sa = orc.scaled_quantize(act)
    c.update({"act.in": gu, "act.out": act, "quant.q": qa, "quant.scale": sa})
    # --- W8A8 GEMM
    a = torch.randint(-127, 128, (8, 256), generator=g, dtype=torch.int8)
    wq = torch.randint(-128, 128, (16, 256), generator=g, dtype=torch.int8)
    a_s = torch.rand(8, generator=g) * 0.05 + 0.01
    w_s = torch.rand(16, generator=g) * 0.02 + 0.01
    bias = torch.randn(16, generator=g).to(bf)
    y, acc = orc.scaled_matmul(a, wq, a_s, w_s, bf, bias, want_acc=True)
    c.update({"gemm.a": a, "gemm.w": wq, "gemm.a_scale": a_s, "gemm.w_scale": w_s, "gemm.bias": bias, "gemm.out": y,
              "gemm.acc": acc})
    # --- KV write + paged decode + prefill + chunked prefill
    nq, nkv, d, bs = 4, 2, 64, 16
    kv_lens, q_lens = [40, 17], [1, 1]
    blocks = [[5, 2, 7], [1, 4]]
    md = orc.build_batch_metadata(kv_lens, q_lens, blocks, bs)
    kc = torch.randn(9, bs, nkv, d, generator=g).to(bf)
    vc = torch.randn(9, bs, nkv, d, generator=g).to(bf)
    knew = torch.randn(2, nkv, d, generator=g).to(bf)
    vnew = torch.randn(2, nkv, d, generator=g).to(bf)
    kc2, vc2 = kc.clone(), vc.clone()
    orc.reshape_paged_cache(md["new_cache_slots"], knew, vnew, kc2, vc2)
    qd = torch.randn(2, nq, d, generator=g).to(bf)
    scale = 1.0 / math.sqrt(d)
    od = orc.paged_attention(qd, kc2, vc2, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale)
    c.update({"kv.kc_in": kc, "kv.vc_in": vc, "kv.k": knew, "kv.v": vnew, "kv.slots": md["new_cache_slots"],
              "kv.kc_out": kc2, "kv.vc_out": vc2, "decode.q": qd, "decode.kv_lens": md["kv_seq_lens"],
              "decode.block_table": md["block_tables"], "decode.out": od})
    lens = [33, 20]
    T = sum(lens)
    qp = torch.randn(T, nq, d, generator=g).to(bf)
    kp = torch.randn(T, nkv, d, generator=g).to(bf)
    vp = torch.randn(T, nkv, d, generator=g).to(bf)
    cu = torch.tensor([0, 33, 53], dtype=torch.int32)
    op = orc.attention_varlen(qp, kp, vp, cu, cu, scale, causal=True)
    c.update({"prefill.q": qp, "prefill.k": kp, "prefill.v": vp, "prefill.cu": cu, "prefill.out": op})
    md2 = orc.build_batch_metadata([40, 17], [5, 3], blocks, bs)
    qc = torch.randn(8, nq, d, generator=g).to(bf)
    oc = orc.paged_attention(qc, kc2, vc2, md2["q_cu_seq_lens"], md2["kv_seq_lens"], md2["block_tables"], scale, causal=True)
    c.update({"chunked.q": qc, "chunked.cu_q": md2["q_cu_seq_lens"], "chunked.out": oc})
    # --- fp8
    xf = (torch.randn(3, 64, generator=g) * 4).to(bf)
    qf, sf = orc.fp8_scaled_quantize(xf)
    c.update({"fp8.x": xf, "fp8.q": qf, "fp8.scale": sf})
    # --- MoE index / gating, sampler
    eid = torch.randint(0, 8, (10, 2), generator=g, dtype=torch.int32)
    for kname, t in zip(("src_dst", "dst_src", "sizes"), orc.moe_compute_index(eid, 8)[:3]):
        c["moe_index." + kname] = t
    c["moe_index.expert_id"] = eid
    gate = torch.randn(6, 16, generator=g)
    tw, ti = orc.moe_fused_topk(gate, 4, True, None, "softmax")
    c.update({"topk.gating": gate, "topk.weights": tw, "topk.ids": ti})
    c["philox.u_seed7_off5"] = orc.philox_uniform(8, 7, 5)
    draft = torch.tensor([1, 2, 3, 0], dtype=torch.int32)
    n = torch.tensor([2, 2], dtype=torch.int32)
    cun = torch.tensor([2, 4], dtype=torch.int32)
    dp = torch.softmax(torch.randn(4, 9, generator=g), -1)
    tp = torch.softmax(torch.randn(4, 9, generator=g), -1)
    ur, up = torch.rand(4, generator=g), torch.rand(4, 9, generator=g)
    bonus = torch.tensor([5, 6], dtype=torch.int32)
    c.update({"rej.draft": draft, "rej.n": n, "rej.cu": cun, "rej.dp": dp, "rej.tp": tp, "rej.ur": ur, "rej.up": up,
              "rej.bonus": bonus, "rej.out": orc.rejection_sample(draft, n, cun, dp, tp, bonus, ur, up)})
    # --- (appended; the generator state above is untouched) MLA decode / prefill over a paged latent cache, page = 64
    Hm, bsm = 16, 64
    m_kv, m_q = [70, 130], [3, 6]
    m_blocks = [[4, 1], [0, 5, 2]]
    mdm = orc.build_batch_metadata(m_kv, m_q, m_blocks, bsm)
    lat = (torch.randn(6, bsm, 1, 576, generator=g) * 0.5).to(bf)
    qm = torch.randn(sum(m_q), Hm, 576, generator=g).to(bf)
    m_scale = 192 ** -0.5
    om = orc.paged_attention(qm, lat, lat, mdm["q_cu_seq_lens"], mdm["kv_seq_lens"], mdm["block_tables"], m_scale,
                             causal=True, dv=512)
    last = mdm["q_cu_seq_lens"][1:].long() - 1
    c.update({"mla.cache": lat, "mla.q": qm, "mla.cu_q": mdm["q_cu_seq_lens"], "mla.kv_lens": mdm["kv_seq_lens"],
              "mla.block_table": mdm["block_tables"], "mla.prefill_out": om, "mla.decode_out": om[last].contiguous()})
    # --- grouped gating top-k (DeepSeek-V3 style: sigmoid + bias, 4 groups, 2 kept) and the softmax form
    gg = torch.randn(7, 32, generator=g) * 2
    gb = torch.randn(32, generator=g) * 0.1
    w1, i1 = orc.moe_grouped_topk(gg, 4, 4, 2, True, gb, "sigmoid", 2.5)
    w2, i2 = orc.moe_grouped_topk(gg, 3, 8, 3, False, None, "softmax", 1.0)
    c.update({"gtopk.gating": gg, "gtopk.bias": gb, "gtopk.sig_w": w1, "gtopk.sig_ids": i1, "gtopk.soft_w": w2,
              "gtopk.soft_ids": i2})
    # --- MoE combine and the (bf16) grouped GEMM
    g2 = torch.randn(10 * 2, 64, generator=g).to(bf)
    cw = torch.rand(10, 2, generator=g)
    c.update({"combine.gemm2": g2, "combine.w": cw, "combine.out": orc.moe_combine(g2, cw, 10, 2)})
    xa = torch.randn(20, 64, generator=g).to(bf)
    ww = (torch.randn(8, 48, 64, generator=g) / 8).to(bf)
    c.update({"ggemm.a": xa, "ggemm.w": ww, "ggemm.out": orc.group_gemm(xa, ww, c["moe_index.sizes"])})
    return c


def main():
    arrs = {}
    for k, t in cases().items():
        if t.dtype in (torch.bfloat16, torch.float16):
            arrs[k] = t.contiguous().view(torch.int16).numpy()
            arrs[k + "__dtype"] = np.array(str(t.dtype))
        else:
            arrs[k] = t.contiguous().numpy()
    np.savez_compressed(OUT, **arrs)
    print(OUT, sum(a.nbytes for a in arrs.values()), "bytes raw")


if __name__ == "__main__":
    main()
