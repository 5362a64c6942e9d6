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
