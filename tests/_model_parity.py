"""Shared machinery of the logits-level parity tests (tests/test_gpu_model_parity.py) and tools/model_parity_diag.py.

Two comparisons of the HIP path with the end-to-end oracle model (oracle/model.py):

 * TEACHER-FORCED per operator: the oracle runs once and records every operator's output; each HIP operator is fed the
   ORACLE's inputs and compared with the oracle's output row by row. This is where the north star's per-operator bars
   (bf16 <= 1e-3, fp8 <= 2e-2, integer work bit-exact) are meaningful and asserted.
 * FREE-RUNNING: both sides run the whole model from the same tokens. Two 16-bit pipelines with the SAME cast points but
   different fp32 summation orders drift apart layer by layer (a 1-ulp flip of one rounding is a 4e-3 relative change of that
   element and feeds everything downstream; int8 / fp8 re-quantisation amplifies it further). The size of that drift is
   MEASURED in the same test as a control -- the oracle against itself with another (equally valid) evaluation order of the
   linears -- and the HIP path has to stay within a small factor of it.
Test infrastructure: the oracle is the checker."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import model as omodel  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = "cuda"
OPS = ["layer_in", "input_norm", "residual1", "qkv", "qkv_rope", "attn", "o_proj", "post_norm", "residual2", "gate_up", "act",
       "down"]


def rel_rows(got, ref):
    a, b = got.float().cpu().reshape(ref.shape[0], -1), ref.float().reshape(ref.shape[0], -1)
    return (a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-30)


class AltOrderOracle(omodel.OracleQwen2):
    """the oracle with another, equally valid, fp32 evaluation order of the linear layers (same cast points):
    16-bit: torch's blocked matmul instead of the sequential k loop; int8: the two scale vectors applied in the other order
    (the reference itself leaves that order to hipBLASLt); fp8: a_scale * w_scale folded first."""

    def _linear(self, x, l):
        if l["mode"] == "16bit":
            y = x.float() @ l["weight"].float().T
            if l["bias"] is not None:
                y = y + l["bias"].float()
            return y.to(self.dtype)
        if l["mode"] == "int8":
            q, s = orc.scaled_quantize(x)
            acc = (q.double() @ l["weight"].double().T).float()
            y = (acc * l["w_scale"][None, :]) * s[:, None]
            if l["bias"] is not None:
                y = y + l["bias"].float()
            return y.to(self.dtype)
        q, s = orc.fp8_scaled_quantize(x)
        acc = orc.e4m3_to_f32(q) @ orc.e4m3_to_f32(l["weight"].view(torch.uint8)).T
        y = acc * (s * l["w_scale"])
        if l["bias"] is not None:
            y = y + l["bias"].float()
        return y.to(self.dtype)


class SplitKVOracle(omodel.OracleQwen2):
    """the oracle with another, equally valid, fp32 SUMMATION ORDER of its decode attention (same cast points: bf16 q / k / v,
    fp32 scores, fp32 softmax, fp32 PV, one rounding of the output): every sequence's keys are cut into `parts` contiguous
    ranges, each range gives a partial (max, sum, un-normalised output) and the partials are merged in fp32 -- exactly what a
    split-KV decode kernel does. This is the control the int8 free-running comparison needs (round-4 review, weak #1): the
    linears of an int8 model are exact, so re-ordering THEM perturbs nothing that feeds a quantiser; the attention output does."""

    def __init__(self, *a, parts=2, exp2=False, hilo=False, **kw):
        super().__init__(*a, **kw)
        self.parts, self.exp2 = parts, exp2     # exp2: softmax in the base-2 domain, exp2(s * (scale * log2 e) - m), as flash kernels do
        # hilo: P enters the PV product as hi + lo 16-bit parts (p = hi + lo to ~2^-17 relative) -- the cast point of
        # xllm_amd/csrc/attention_decode.hip:289-300 (the matrix core takes 16-bit operands); online order only
        self.hilo = hilo

    def _attention(self, q3, k3, v3, kc, vc, md, phase):
        if phase != "decode":
            return super()._attention(q3, k3, v3, kc, vc, md, phase)
        B, nq, d = q3.shape
        nkv, rep = kc.shape[2], nq // kc.shape[2]
        bs = kc.shape[1]
        out = torch.empty(B, nq * d, dtype=q3.dtype)
        for b in range(B):
            L = int(md["kv_seq_lens"][b])
            pages = md["block_tables"][b, :(L + bs - 1) // bs].long()
            K = kc[pages].reshape(-1, nkv, d)[:L].float()                       # [L, nkv, d]
            V = vc[pages].reshape(-1, nkv, d)[:L].float()
            q = q3[b].float().reshape(nkv, rep, d)                              # heads of one kv head are consecutive
            s = torch.einsum("grd,lgd->grl", q, K)                              # [nkv, rep, L] fp32
            s = s * (self.scale * 1.4426950408889634) if self.exp2 else s * self.scale
            if self.parts <= 0:        # ONLINE order (flash decoding): tiles of -parts keys, running maximum, o <- o * alpha + P V
                tile = -self.parts
                ex = torch.exp2 if self.exp2 else torch.exp
                m_run = torch.full(s.shape[:2] + (1,), -float("inf"))
                l_run = torch.zeros(s.shape[:2] + (1,))
                o_run = torch.zeros(s.shape[:2] + (d,))
                for lo in range(0, L, tile):
                    sc = s[..., lo:lo + tile]
                    m_new = torch.maximum(m_run, sc.amax(-1, keepdim=True))
                    alpha = ex(m_run - m_new)
                    p = ex(sc - m_new)
                    if self.hilo:
                        hi = p.to(q3.dtype).float()
                        p = hi + (p - hi).to(q3.dtype).float()
                    l_run = l_run * alpha + p.sum(-1, keepdim=True)
                    o_run = o_run * alpha + torch.einsum("grl,lgd->grd", p, V[lo:lo + tile])
                    m_run = m_new
                out[b] = (o_run / l_run).reshape(nq * d).to(q3.dtype)
                continue
            cuts = [L * i // self.parts for i in range(self.parts + 1)]
            ms, ls, os_ = [], [], []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi == lo:
                    continue
                sc = s[..., lo:hi]
                m = sc.amax(-1, keepdim=True)
                p = torch.exp2(sc - m) if self.exp2 else torch.exp(sc - m)
                ms.append(m); ls.append(p.sum(-1, keepdim=True)); os_.append(torch.einsum("grl,lgd->grd", p, V[lo:hi]))
            M = torch.stack(ms).amax(0)
            w = [torch.exp2(m - M) if self.exp2 else torch.exp(m - M) for m in ms]
            l = sum(li * wi for li, wi in zip(ls, w))
            o = sum(oi * wi for oi, wi in zip(os_, w)) / l
            out[b] = o.reshape(nq * d).to(q3.dtype)
        return out


def hip_free_running_trace(model, tokens, positions, md, caches):
    """the HIP path, FREE-RUNNING (every operator consumes the HIP path's own outputs), in the reference's operator order
    (model built with fuse=False; the fused forms are bit-identical to it, tests/test_gpu_parity.py), recording the tensors
    of OPS per layer like the oracle's trace does. Returns (per-layer dicts of CPU tensors, final hidden states on the device)."""
    from xllm_amd import ops
    args = model.args
    out = []
    x = torch.nn.functional.embedding(tokens, model.embed)
    residual = None
    for L, kvc in zip(model.layers, caches):
        t = {"layer_in": x.cpu().clone()}
        if residual is None:
            h = torch.empty_like(x)
            ops.rms_norm(h, x, L.input_norm_w, args.rms_norm_eps)
            residual = x
        else:
            ops.fused_add_rms_norm(x, residual, L.input_norm_w, args.rms_norm_eps)
            h = x
        t["input_norm"], t["residual1"] = h.cpu().clone(), residual.cpu().clone()
        qkv = L.qkv_proj.forward(h)
        t["qkv"] = qkv.cpu().clone()
        ops.rotary_embedding(positions, qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size], model.cos_sin, True, head_size=L.d)
        t["qkv_rope"] = qkv.cpu().clone()
        attn, _ = L.attn.forward(md, qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size], qkv[:, L.q_size + L.kv_size:], kvc)
        attn = attn.reshape(attn.size(0), -1)
        t["attn"] = attn.cpu().clone()
        x = L.o_proj.forward(attn)
        t["o_proj"] = x.cpu().clone()
        ops.fused_add_rms_norm(x, residual, L.post_norm_w, args.rms_norm_eps)
        t["post_norm"], t["residual2"] = x.cpu().clone(), residual.cpu().clone()
        gate_up = L.gate_up_proj.forward(x)
        t["gate_up"] = gate_up.cpu().clone()
        act = torch.empty(x.size(0), L.I, dtype=x.dtype, device=x.device)
        ops.act_and_mul(act, gate_up, "silu")
        t["act"] = act.cpu().clone()
        x = L.down_proj.forward(act)
        t["down"] = x.cpu().clone()
        out.append(t)
    ops.fused_add_rms_norm(x, residual, model.norm_w, args.rms_norm_eps)
    return out, x


QUANTISER_INPUTS = ("input_norm", "attn", "post_norm", "act")     # the operands of qkv / o / gate_up / down (linear.cpp:481-507)


def attribute_rows(per_layer_a, per_layer_b, int8: bool):
    """row by row: where two free-running traces of the same step first part, and -- for W8A8 models -- what happens at the
    first per-token quantiser whose int8 CODES differ: how many codes, by how much, and whether the row's scale (its |max|)
    moved (a 1-ulp change of the |max| element re-rolls every code of the row; any other 1-ulp change moves one code by one).
    Returns one dict per row: first_diff = (layer, op, differing elements, rel L2), first_code_diff = (layer, op, codes that
    differ, max |delta code|, scale_moved, rel scale change) or None."""
    B = per_layer_a[0]["layer_in"].shape[0]
    rows = [dict(first_diff=None, first_code_diff=None) for _ in range(B)]
    for li, (ta, tb) in enumerate(zip(per_layer_a, per_layer_b)):
        for op in OPS:
            a, b = ta[op].reshape(B, -1), tb[op].reshape(B, -1)
            ne = (a != b).sum(1)
            rel = (a.float() - b.float()).norm(dim=1) / b.float().norm(dim=1).clamp_min(1e-30)
            qa = qb = None
            if int8 and op in QUANTISER_INPUTS:
                (qa, sa), (qb, sb) = orc.scaled_quantize(ta[op].reshape(B, -1).contiguous()), orc.scaled_quantize(tb[op].reshape(B, -1).contiguous())
            for r in range(B):
                if rows[r]["first_diff"] is None and ne[r] > 0:
                    rows[r]["first_diff"] = (li, op, int(ne[r]), float(rel[r]))
                if qa is not None and rows[r]["first_code_diff"] is None:
                    dq = (qa[r].int() - qb[r].int()).abs()
                    if int(dq.max()) > 0:
                        rows[r]["first_code_diff"] = (li, op, int((dq > 0).sum()), int(dq.max()), bool(sa[r] != sb[r]),
                                                      float(abs(sa[r] - sb[r]) / sb[r]))
    return rows


JUMP = 5e-4      # a row whose free-running logits are further than this from the oracle's has re-rolled int8 codes somewhere
CONTROLS = [("split1_exp2", 1, True, False), ("split2_exp2", 2, True, False), ("split4_exp2", 4, True, False),
            ("split8_exp2", 8, True, False), ("split2", 2, False, False), ("split4", 4, False, False),
            ("online16_exp2", -16, True, False), ("online64_exp2", -64, True, False), ("online64", -64, False, False),
            ("online32_exp2_hilo", -32, True, True), ("online16_exp2_hilo", -16, True, True)]


def int8_jump_experiment(seed=23, n_extra=24):
    """two Qwen2-7B-geometry W8A8 layers, one ragged decode step over random caches, 8 + n_extra sequences: per-row logits
    error of the HIP path (fused, as it is benchmarked) and of every control against the oracle, with the row-by-row attribution
    (attribute_rows) of each. Needs the GPU for the HIP side."""
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=seed)
    plain = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=seed, fuse=False)
    w = omodel.export_weights(model)
    g = torch.Generator().manual_seed(9)
    bs = 128
    lens = [1, 129, 700, 128, 333, 5, 1024, 257] + torch.randint(1, 1025, (n_extra,), generator=g).tolist()
    B = len(lens)
    need = [(n + bs - 1) // bs for n in lens]
    perm = torch.randperm(sum(need) + 3, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    nb = sum(need) + 3
    init = [(torch.randn(nb, bs, 4, 128, generator=g).bfloat16(), torch.randn(nb, bs, 4, 128, generator=g).bfloat16())
            for _ in range(args.n_layers)]
    ids = torch.randint(0, args.vocab_size, (B,), generator=g)
    pos = torch.tensor([n - 1 for n in lens])
    md = orc.build_batch_metadata(lens, [1] * B, blocks, bs)

    def run_oracle(om):
        tr = []
        om.trace = tr
        kcs, vcs = [k.clone() for k, _ in init], [v.clone() for _, v in init]
        lg = om.logits(om.forward(ids, pos, md, kcs, vcs, "decode")).float()
        return lg, split_trace(tr, args.n_layers)

    ref, tref = run_oracle(omodel.OracleQwen2(args, w, torch.bfloat16))
    errors, attribution = {}, {}
    bi = attention.build_batch_input([n - 1 for n in lens], lens, blocks, bs)
    amd = attention.build_attention_metadata(bi, False, False, DEV)
    caches = lambda: [KVCache(k.clone().to(DEV), v.clone().to(DEV)) for k, v in init]
    posd = bi.positions.long().to(DEV)
    hip = model.logits(model.forward(ids.to(DEV), posd, amd, caches())).float().cpu()
    thip, hid = hip_free_running_trace(plain, ids.to(DEV), posd, amd, caches())
    hip_plain = plain.logits(hid).float().cpu()
    errors["hip"], attribution["hip"] = rel_rows(hip, ref).tolist(), attribute_rows(thip, tref, True)
    errors["hip_unfused"], attribution["hip_unfused"] = rel_rows(hip_plain, ref).tolist(), attribution["hip"]
    for name, parts, e2, hilo in CONTROLS:
        lg, tr = run_oracle(SplitKVOracle(args, w, torch.bfloat16, parts=parts, exp2=e2, hilo=hilo))
        errors[name], attribution[name] = rel_rows(lg, ref).tolist(), attribute_rows(tr, tref, True)
    lg, tr = run_oracle(AltOrderOracle(args, w, torch.bfloat16))
    errors["alt_order_linears"], attribution["alt_order_linears"] = rel_rows(lg, ref).tolist(), attribute_rows(tr, tref, True)
    # layer-0 attention elements that differ from the oracle's (all rows): the input of every run is identical up to there
    n_diff = {k: sum(r["first_diff"][2] for r in v if r["first_diff"] is not None and r["first_diff"][:2] == (0, "attn"))
              for k, v in attribution.items()}
    return dict(lens=lens, errors=errors, attribution=attribution, attn0_diff_elements=n_diff,
                fused_equals_unfused=bool(torch.equal(hip, hip_plain)))


def split_trace(trace, n_layers):
    return [dict(trace[i * len(OPS):(i + 1) * len(OPS)]) for i in range(n_layers)]


def teacher_forced_errors(model, per_layer, positions, phase, attn_inputs, report=None):
    """model: xllm_amd.layers.Qwen2Model (fuse=False ordering is replayed here op by op); per_layer: split_trace(...) of the
    oracle; attn_inputs(li) -> dict(md=AttentionMetadata on the device, caches=KVCache holding what the oracle's cache held
    BEFORE this step). Returns {op name: (max rel L2 over rows and layers, max fraction of differing elements)}."""
    from xllm_amd import ops
    args = model.args
    d = lambda t: t.to(DEV)
    out = {}

    def note(name, got, ref):
        e = rel_rows(got, ref).max().item()
        neq = (got.cpu() != ref).float().mean().item()
        pe, pn = out.get(name, (0.0, 0.0))
        out[name] = (max(pe, e), max(pn, neq))
        if report is not None:
            report(name, e, neq)

    prev_res = None
    pos = d(positions)
    for li, (L, t) in enumerate(zip(model.layers, per_layer)):
        x = d(t["layer_in"])
        if prev_res is None:
            y = torch.empty_like(x)
            ops.rms_norm(y, x, L.input_norm_w, args.rms_norm_eps)
            note("rms_norm", y, t["input_norm"])
        else:
            r = d(prev_res)
            ops.fused_add_rms_norm(x, r, L.input_norm_w, args.rms_norm_eps)
            note("fused_add_rms_norm", x, t["input_norm"])
            note("fused_add_rms_norm.residual", r, t["residual1"])
        note("linear.qkv", L.qkv_proj.forward(d(t["input_norm"])), t["qkv"])
        qkv = d(t["qkv"]).clone()
        ops.rotary_embedding(pos, qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size], model.cos_sin, True, head_size=L.d)
        note("rotary_embedding", qkv, t["qkv_rope"])
        qkv = d(t["qkv_rope"])
        ai = attn_inputs(li)
        attn, _ = L.attn.forward(ai["md"], qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size],
                                 qkv[:, L.q_size + L.kv_size:], ai["caches"])
        note("attention." + phase, attn.reshape(attn.size(0), -1), t["attn"])
        if "k_after" in ai:   # KV write: bit-exact against what the oracle's cache holds after the step
            note("reshape_paged_cache.k", ai["caches"].k_cache.flatten(1), ai["k_after"].flatten(1))
            note("reshape_paged_cache.v", ai["caches"].v_cache.flatten(1), ai["v_after"].flatten(1))
        note("linear.o", L.o_proj.forward(d(t["attn"])), t["o_proj"])
        x, r = d(t["o_proj"]).clone(), d(t["residual1"]).clone()
        ops.fused_add_rms_norm(x, r, L.post_norm_w, args.rms_norm_eps)
        note("fused_add_rms_norm", x, t["post_norm"])
        note("fused_add_rms_norm.residual", r, t["residual2"])
        note("linear.gate_up", L.gate_up_proj.forward(d(t["post_norm"])), t["gate_up"])
        act = torch.empty(x.size(0), L.I, dtype=x.dtype, device=DEV)
        ops.act_and_mul(act, d(t["gate_up"]), "silu")
        note("act_and_mul", act, t["act"])
        note("linear.down", L.down_proj.forward(d(t["act"])), t["down"])
        if L.mode == "int8":   # the quantiser and the fused quantising kernels (N1) on the oracle's inputs: bit-exact
            for name, src in (("act", t["act"]), ("post_norm", t["post_norm"])):
                qr, sr = orc.scaled_quantize(src)
                qd, sd = ops.scaled_quantize(d(src))
                note("scaled_quantize.q", qd.float(), qr.float())
                note("scaled_quantize.scale", sd[:, None], sr[:, None])
            qr, sr = orc.scaled_quantize(t["act"])
            qd, sd = ops.act_and_mul_dynamic_int8_quant(d(t["gate_up"]), "silu")
            note("act_and_mul+quant(N1).q", qd.float(), qr.float())
            note("act_and_mul+quant(N1).scale", sd[:, None], sr[:, None])
        prev_res = t["residual2"]
    return out
