"""TEST INFRASTRUCTURE ONLY (never imported by xllm_amd/): CPU restatement of the reference's MLA attention layer in the ABSORBED
form it runs on its CUDA / DCU backends, composed from the oracle's operators.

Follows DeepseekV2AttentionImpl (layers/dcu/deepseek_v2_attention.cpp): prepare_query :156-168 (q_a_proj -> RMSNorm -> q_b_proj,
split into q_nope | q_pe), the latent row [rms_norm(c_kv) (kv_lora) || rope(k_pe) (rope)] written to the paged cache
(store_latent_cache :170-178), q_nope absorbed by w_kc (the K half of kv_b_proj, load_state_dict :335-339), scores over the
(kv_lora + rope)-dim latent with scale (nope + rope)^-0.5 * mscale^2 (:148-154), values = the first kv_lora dims, then
bmm(w_vc) (project_output :180-187) and o_proj. The rotary halves are in DeepSeek's layout: the interleaved pairs of the
checkpoint are de-interleaved first ([.., d/2, 2] -> [.., 2, d/2], :35-46) and rotated neox-style.

Pinned (tests/test_oracle_mla.py) against an independent implementation: HuggingFace `DeepseekV3Attention` (transformers), which
computes the same layer in the NON-absorbed form (expands the latent through kv_b_proj into per-head keys and values) -- the two
agree to fp32 rounding on prefill and on decode over a cache, which pins the absorption, the rope layout, the scale and the
causal alignment of this restatement. Parity pinned: yes (HF), not by a golden of the reference itself (it has none for the
DCU path: tests/core/layers/mlu/mla_test.cpp exercises the MLU indexer variant).
"""
import math

import torch

from . import oracle as orc


def to_deepseek_rope_layout(t: torch.Tensor) -> torch.Tensor:
    """deepseek_v2_attention.cpp:35-46: [.., d] viewed as [.., d/2, 2], transposed to [.., 2, d/2] (evens first)"""
    shape = t.shape
    return t.reshape(*shape[:-1], shape[-1] // 2, 2).transpose(-1, -2).reshape(shape).contiguous()


class OracleMLA:
    def __init__(self, weights: dict, n_heads: int, q_lora: int, kv_lora: int, nope: int, rope: int, v_dim: int, eps: float,
                 rope_theta: float, max_pos: int, dtype=torch.float32, mscale: float = 1.0):
        """weights: q_a_proj [q_lora, H], q_a_layernorm [q_lora], q_b_proj [h*(nope+rope), q_lora], kv_a_proj_with_mqa
        [kv_lora+rope, H], kv_a_layernorm [kv_lora], kv_b_proj [h*(nope+v), kv_lora], o_proj [H, h*v] (checkpoint names)"""
        self.h, self.q_lora, self.kv_lora, self.nope, self.rope, self.v = n_heads, q_lora, kv_lora, nope, rope, v_dim
        self.eps, self.dtype = eps, dtype
        w = {k: v.to(dtype).contiguous() for k, v in weights.items()}
        self.w = w
        kv_b = w["kv_b_proj"].unflatten(0, (n_heads, nope + v_dim))
        self.w_kc = kv_b[:, :nope].contiguous()                       # [h, nope, kv_lora]
        self.w_vc = kv_b[:, nope:].transpose(1, 2).contiguous()       # [h, kv_lora, v]
        self.scale = float((nope + rope) ** -0.5) * mscale * mscale
        self.cos_sin = orc.build_cos_sin_cache(max_pos, rope, rope_theta, dtype)

    def _norm(self, x, w):
        out = torch.empty_like(x)
        orc.rms_norm(out, x.contiguous(), w, self.eps)
        return out

    def forward(self, positions, hidden, md: dict, latent_cache, causal: bool):
        """hidden [T, H]; md = orc.build_batch_metadata(...); latent_cache [n_blocks, bs, 1, kv_lora + rope] (updated in place)"""
        T = hidden.size(0)
        x = hidden.to(self.dtype)
        latent = orc.matmul(x, self.w["kv_a_proj_with_mqa"])
        c_kv = self._norm(latent[:, :self.kv_lora], self.w["kv_a_layernorm"])
        k_pe = to_deepseek_rope_layout(latent[:, self.kv_lora:].contiguous())
        orc.rotary_embedding(positions, k_pe, None, self.cos_sin, self.rope, True)
        row = torch.cat([c_kv, k_pe], -1).contiguous()
        orc.reshape_paged_cache(md["new_cache_slots"], row.view(T, 1, -1), None, latent_cache, None)
        q = orc.matmul(self._norm(orc.matmul(x, self.w["q_a_proj"]), self.w["q_a_layernorm"]), self.w["q_b_proj"])
        q = q.view(T, self.h, self.nope + self.rope)
        q_pe = to_deepseek_rope_layout(q[..., self.nope:].contiguous()).view(T, self.h * self.rope)
        orc.rotary_embedding(positions, q_pe, None, self.cos_sin, self.rope, True)
        q_abs = torch.einsum("thn,hnk->thk", q[..., :self.nope].float(), self.w_kc.float()).to(self.dtype)
        q_in = torch.cat([q_abs, q_pe.view(T, self.h, self.rope)], -1).contiguous()
        attn = orc.paged_attention(q_in, latent_cache, latent_cache, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"],
                                   self.scale, causal=causal, dv=self.kv_lora)            # [T, h, kv_lora]
        out = torch.einsum("thk,hkv->thv", attn.view(T, self.h, self.kv_lora).float(), self.w_vc.float()).to(self.dtype)
        return orc.matmul(out.reshape(T, self.h * self.v).contiguous(), self.w["o_proj"])
