"""End-to-end CPU restatement of the reference's Qwen2 forward over the oracle operators.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by xllm_amd/.

Operator order (each step is one oracle operator of oracle.py, i.e. one xllm::kernel::* call of the reference):
  LlmModelImplBase::forward            xllm/models/llm/llm_model_base.h:60-125   embed -> layers -> norm_(h, residual)
  LlmModelImplBase::logits             llm_model_base.h:193-204                  lm_head(hidden[selected])
  Qwen2DecoderLayerImpl::forward       xllm/core/layers/qwen2_decoder_layer.cpp:87-110 (apply_norm :66-85)
  Qwen2AttentionImpl::forward          layers/common/qwen2_attention.cpp:132-193 qkv -> rope -> KV write + attention -> o
  DenseMLPImpl::forward                layers/common/dense_mlp.cpp:97-116        gate_up -> act_and_mul -> down
  linear, by quantisation              layers/common/linear.cpp:137-182 (fp8: per-tensor dynamic quant + scaled mm),
                                       :481-507 (w8a8-dynamic: per-token int8 quant + scaled_matmul), F::linear else;
                                       lm_head is never quantised (:512-520)
  AttentionImpl::forward               layers/dcu/flash_attention.cpp:291-376    KV write, then prefill / chunked / decode

BASELINE.json config 1 (Qwen2-0.5B fp32, bs=1, ctx=128, greedy decode "on the reference CPU path") is `greedy_generate`
with dtype=float32, mode="16bit": the reference has no CPU backend, so this composition of the restated
operators IS the CPU path the metric is quoted on.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import oracle as orc


def export_weights(model) -> Dict[str, object]:
    """CPU copies of an xllm_amd.layers.Qwen2Model's (TP=1) parameters -- the two sides run on identical numbers."""
    assert model.tp is None or model.tp.world_size() == 1

    def lin(l):
        d = dict(mode=l.mode, weight=l.weight.cpu(), bias=None if l.bias is None else l.bias.cpu())
        if l.mode in ("int8", "fp8"):
            d["w_scale"] = l.w_scale.float().cpu()
        return d

    return dict(
        embed=model.embed.cpu(), norm_w=model.norm_w.cpu(), lm_head=lin(model.lm_head), cos_sin=model.cos_sin.cpu(),
        layers=[dict(input_norm_w=l.input_norm_w.cpu(), post_norm_w=l.post_norm_w.cpu(), qkv=lin(l.qkv_proj),
                     o=lin(l.o_proj), gate_up=lin(l.gate_up_proj), down=lin(l.down_proj)) for l in model.layers])


class OracleQwen2:
    def __init__(self, args, weights: Dict[str, object], dtype=torch.bfloat16, p_round: bool = False, trace=None):
        self.args, self.w, self.dtype, self.p_round = args, weights, dtype, p_round
        self.trace = trace    # optional list: (name, tensor copy) of every operator output, in order (diagnostics)
        self.nq, self.nkv, self.d = args.n_heads, args.n_kv_heads, args.head_dim
        self.q_size, self.kv_size = self.nq * self.d, self.nkv * self.d
        self.scale = math.sqrt(1.0 / self.d)          # qwen2_attention.cpp:70

    def _t(self, name, t):
        if self.trace is not None:
            self.trace.append((name, t.clone()))
        return t

    # ---- linear.cpp ------------------------------------------------------------------------------------------------
    def _linear(self, x, l):
        if l["mode"] == "int8":                                              # linear.cpp:481-507
            q, s = orc.scaled_quantize(x)
            return orc.scaled_matmul(q, l["weight"], s, l["w_scale"], self.dtype, l["bias"])
        if l["mode"] == "fp8":                                               # linear.cpp:137-182
            q, s = orc.fp8_scaled_quantize(x)
            return orc.fp8_scaled_matmul(q, l["weight"].view(torch.uint8), s, l["w_scale"], self.dtype, l["bias"])
        return orc.matmul(x, l["weight"], l["bias"])                         # F::linear

    def _apply_norm(self, x, residual, w):                                   # qwen2_decoder_layer.cpp:66-85
        eps = self.args.rms_norm_eps
        if residual is None:
            out = torch.empty_like(x)
            orc.rms_norm(out, x, w, eps)
            return out, x
        orc.fused_add_rms_norm(x, residual, w, eps)                          # both updated in place
        return x, residual

    def _attention(self, q3, k3, v3, kc, vc, md, phase):                     # flash_attention.cpp:320-376
        if phase == "prefill":
            return orc.attention_varlen(q3, k3, v3, md["q_cu_seq_lens"], md["kv_cu_seq_lens"], self.scale, True, -1,
                                        self.p_round)
        return orc.paged_attention(q3, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"],
                                   self.scale, phase == "chunked", -1, self.p_round)

    # ---- one decoder layer -----------------------------------------------------------------------------------------
    def _layer(self, lw, x, residual, positions, md, kc, vc, phase):
        t = self._t
        t("layer_in", x)
        x, residual = self._apply_norm(x, residual, lw["input_norm_w"])
        t("input_norm", x); t("residual1", residual)
        qkv = t("qkv", self._linear(x, lw["qkv"]))
        q = qkv[:, :self.q_size]
        k = qkv[:, self.q_size:self.q_size + self.kv_size]
        v = qkv[:, self.q_size + self.kv_size:]
        orc.rotary_embedding(positions, q, k, self.w["cos_sin"], self.d, True)
        t("qkv_rope", qkv)
        T = q.shape[0]
        q3 = q.unflatten(-1, (self.nq, self.d))
        k3 = k.unflatten(-1, (self.nkv, self.d))
        v3 = v.unflatten(-1, (self.nkv, self.d))
        orc.reshape_paged_cache(md["new_cache_slots"], k3, v3, kc, vc)       # flash_attention.cpp:310-318
        attn = t("attn", self._attention(q3, k3, v3, kc, vc, md, phase))
        x = t("o_proj", self._linear(attn.view(T, self.q_size), lw["o"]))
        x, residual = self._apply_norm(x, residual, lw["post_norm_w"])
        t("post_norm", x); t("residual2", residual)
        gate_up = t("gate_up", self._linear(x, lw["gate_up"]))
        act = torch.empty(T, gate_up.shape[1] // 2, dtype=gate_up.dtype)
        orc.act_and_mul(act, gate_up, "silu")
        t("act", act)
        return t("down", self._linear(act, lw["down"])), residual

    # ---- llm_model_base.h ------------------------------------------------------------------------------------------
    def forward(self, tokens, positions, md, k_caches: List[torch.Tensor], v_caches: List[torch.Tensor], phase: str):
        """phase in {"prefill", "chunked", "decode"}; md = oracle.build_batch_metadata(...) dict; returns the hidden
        states after the final norm, [T, H]."""
        assert phase in ("prefill", "chunked", "decode")
        h = self.w["embed"][tokens.long()].to(self.dtype).contiguous()
        residual = None
        positions = positions.to(torch.int64)
        for lw, kc, vc in zip(self.w["layers"], k_caches, v_caches):
            h, residual = self._layer(lw, h, residual, positions, md, kc, vc, phase)
        h, _ = self._apply_norm(h, residual, self.w["norm_w"])
        return h

    def logits(self, hidden):
        return self._linear(hidden.contiguous(), self.w["lm_head"])


def greedy_generate(model: OracleQwen2, prompt: torch.Tensor, n_new: int, block_size: int = 128,
                    block_ids: Optional[List[int]] = None, collect_logits: bool = True):
    """bs = 1: prefill the prompt, then n_new - 1 decode steps, argmax sampling (examples/generate.py with
    temperature 0). Returns (token ids [n_new], [logits per generated token] fp32)."""
    a = model.args
    L = prompt.numel()
    n_pages = (L + n_new + block_size - 1) // block_size
    blocks = list(block_ids) if block_ids is not None else list(range(n_pages))
    nb = max(blocks) + 1
    kcs = [torch.zeros(nb, block_size, a.n_kv_heads, a.head_dim, dtype=model.dtype) for _ in model.w["layers"]]
    vcs = [torch.zeros(nb, block_size, a.n_kv_heads, a.head_dim, dtype=model.dtype) for _ in model.w["layers"]]
    out_tokens, out_logits = [], []
    md = orc.build_batch_metadata([L], [L], [blocks], block_size)
    hidden = model.forward(prompt, torch.arange(L), md, kcs, vcs, "prefill")
    lg = model.logits(hidden[-1:]).float()
    for step in range(n_new):
        tok = int(lg[0].argmax())
        out_tokens.append(tok)
        if collect_logits:
            out_logits.append(lg[0].clone())
        if step + 1 == n_new:
            break
        cur = L + step + 1                                   # sequence length after this decode step
        md = orc.build_batch_metadata([cur], [1], [blocks], block_size)
        hidden = model.forward(torch.tensor([tok]), torch.tensor([cur - 1]), md, kcs, vcs, "decode")
        lg = model.logits(hidden).float()
    return torch.tensor(out_tokens), out_logits
