"""The oracle's absorbed MLA layer (oracle/mla.py = DeepseekV2AttentionImpl restated) against HuggingFace DeepseekV3Attention
(non-absorbed formulation, an independent implementation): prefill of ragged sequences, then decode steps over the paged latent
cache the oracle layer wrote. fp32 on both sides: agreement to rounding pins absorption, rope layout, scale, causal alignment."""
import math

import pytest
import torch

from oracle import mla as omla
from oracle import oracle as orc

transformers = pytest.importorskip("transformers")


def _hf_layer(cfg_kw, seed):
    from transformers.models.deepseek_v3 import modeling_deepseek_v3 as m
    cfg = m.DeepseekV3Config(**cfg_kw)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    attn = m.DeepseekV3Attention(cfg, 0).float().eval()
    for p in attn.parameters():
        torch.nn.init.normal_(p, 0.0, 0.08)
    for n in (attn.q_a_layernorm, attn.kv_a_layernorm):
        n.weight.data = torch.rand_like(n.weight) + 0.5
    rot = m.DeepseekV3RotaryEmbedding(cfg)
    return m, cfg, attn, rot


def _hf_run(attn, rot, x, pos, mask):
    """x [1, S, H]; mask additive [1, 1, S, S] or None"""
    cos, sin = rot(x, pos[None])
    with torch.no_grad():
        out, _ = attn(x, (cos, sin), mask)
    return out[0]


@pytest.mark.parametrize("interleave", [True, False])
def test_absorbed_mla_oracle_equals_hf_deepseek_v3_attention(interleave):
    H, heads, q_lora, kv_lora, nope, rope, v = 256, 4, 96, 64, 32, 16, 24
    cfg_kw = dict(hidden_size=H, num_attention_heads=heads, num_key_value_heads=heads, q_lora_rank=q_lora, kv_lora_rank=kv_lora,
                  qk_nope_head_dim=nope, qk_rope_head_dim=rope, v_head_dim=v, max_position_embeddings=512, rope_theta=10000.0,
                  rope_interleave=interleave, attention_bias=False, num_hidden_layers=1, intermediate_size=64,
                  moe_intermediate_size=32, n_routed_experts=4, num_experts_per_tok=2, n_group=1, topk_group=1, vocab_size=64)
    m, cfg, attn, rot = _hf_layer(cfg_kw, seed=3 + interleave)
    sd = {k: p.detach().clone() for k, p in attn.named_parameters()}
    weights = {"q_a_proj": sd["q_a_proj.weight"], "q_a_layernorm": sd["q_a_layernorm.weight"], "q_b_proj": sd["q_b_proj.weight"],
               "kv_a_proj_with_mqa": sd["kv_a_proj_with_mqa.weight"], "kv_a_layernorm": sd["kv_a_layernorm.weight"],
               "kv_b_proj": sd["kv_b_proj.weight"], "o_proj": sd["o_proj.weight"]}
    if not interleave:
        # a checkpoint whose rope dims are already [evens || odds]: the layer's de-interleave must then NOT be applied; the
        # reference always applies it (its checkpoints are interleaved), so feed it the interleaved equivalent of these weights
        def interleave_rows(w, blocks, off, d):   # rows [off, off+d) of each block: half layout -> interleaved pairs
            w = w.clone()
            for b in range(blocks):
                r0 = b * (w.size(0) // blocks) + off
                half = w[r0:r0 + d].clone()
                w[r0:r0 + d:2] = half[:d // 2]
                w[r0 + 1:r0 + d:2] = half[d // 2:]
            return w
        weights["q_b_proj"] = interleave_rows(weights["q_b_proj"], heads, nope, rope)
        weights["kv_a_proj_with_mqa"] = interleave_rows(weights["kv_a_proj_with_mqa"], 1, kv_lora, rope)
    layer = omla.OracleMLA(weights, heads, q_lora, kv_lora, nope, rope, v, cfg.rms_norm_eps, 10000.0, 512)
    assert abs(layer.scale - attn.scaling) < 1e-7

    bs, lens = 16, [37, 5, 64]
    g = torch.Generator().manual_seed(11)
    need = [(L + 8 + bs - 1) // bs for L in lens]
    perm = torch.randperm(sum(need) + 2, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    cache = torch.zeros(sum(need) + 2, bs, 1, kv_lora + rope)
    xs = [torch.randn(L + 3, H, generator=g) for L in lens]            # 3 extra tokens per sequence for the decode steps

    # prefill: ragged batch through the oracle layer, one sequence at a time through HF
    md = orc.build_batch_metadata(lens, lens, blocks, bs)
    hidden = torch.cat([x[:L] for x, L in zip(xs, lens)])
    pos = torch.cat([torch.arange(L) for L in lens])
    got = layer.forward(pos, hidden, md, cache, causal=True)
    off = 0
    for x, L in zip(xs, lens):
        mask = torch.full((L, L), float("-inf")).triu(1)[None, None]
        ref = _hf_run(attn, rot, x[None, :L], torch.arange(L), mask)
        assert torch.allclose(got[off:off + L], ref, rtol=2e-4, atol=2e-5), (got[off:off + L] - ref).abs().max()
        off += L
    # decode: three steps over the cache the oracle layer wrote; HF recomputes the whole prefix each time
    cur = list(lens)
    for step in range(3):
        new = [c + 1 for c in cur]
        md = orc.build_batch_metadata(new, [1] * len(new), blocks, bs)
        hidden = torch.stack([x[c] for x, c in zip(xs, cur)])
        got = layer.forward(torch.tensor(cur), hidden, md, cache, causal=False)
        for b, (x, c) in enumerate(zip(xs, cur)):
            mask = torch.full((c + 1, c + 1), float("-inf")).triu(1)[None, None]
            ref = _hf_run(attn, rot, x[None, :c + 1], torch.arange(c + 1), mask)[-1]
            assert torch.allclose(got[b], ref, rtol=2e-4, atol=2e-5), (step, b, (got[b] - ref).abs().max())
        cur = new
