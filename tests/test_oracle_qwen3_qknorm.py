"""a10: the oracle's fused per-head RMSNorm(q), RMSNorm(k) + RoPE inside the packed qkv (cuda::fused_qk_norm_rope,
kernels/cuda/fused_qknorm_rope.cu:88-..., called from qwen2_attention.cpp:146-171 for Qwen3) against HuggingFace Qwen3-MoE's
attention front end (q_norm / k_norm modules + apply_rotary_pos_emb), an independent implementation. fp32."""
import pytest
import torch

from oracle import oracle as orc

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("nq,nk,d", [(4, 1, 128), (8, 2, 64)])
def test_fused_qk_norm_rope_oracle_equals_hf_qwen3(nq, nk, d):
    from transformers.models.qwen3_moe import modeling_qwen3_moe as q
    H, T, theta = 96, 29, 10000.0
    cfg = q.Qwen3MoeConfig(hidden_size=H, num_attention_heads=nq, num_key_value_heads=nk, head_dim=d, rope_theta=theta,
                           max_position_embeddings=512, num_hidden_layers=1, num_experts=4, num_experts_per_tok=2,
                           moe_intermediate_size=16, intermediate_size=32, vocab_size=64, attention_bias=False)
    torch.manual_seed(nq)
    attn = q.Qwen3MoeAttention(cfg, 0).float().eval()
    attn.q_norm.weight.data = torch.rand(d) + 0.5
    attn.k_norm.weight.data = torch.rand(d) + 0.5
    rot = q.Qwen3MoeRotaryEmbedding(cfg)
    x = torch.randn(1, T, H)
    pos = torch.randint(0, 400, (T,))
    with torch.no_grad():
        qh = attn.q_norm(attn.q_proj(x).view(1, T, nq, d)).transpose(1, 2)
        kh = attn.k_norm(attn.k_proj(x).view(1, T, nk, d)).transpose(1, 2)
        cos, sin = rot(x, pos[None])
        q_ref, k_ref = q.apply_rotary_pos_emb(qh, kh, cos, sin)
        v_ref = attn.v_proj(x)[0]
        qkv = torch.cat([attn.q_proj(x)[0], attn.k_proj(x)[0], v_ref], -1).contiguous()
    cache = orc.build_cos_sin_cache(512, d, theta, torch.float32)
    orc.fused_qk_norm_rope(qkv, nq, nk, nk, d, cfg.rms_norm_eps, attn.q_norm.weight.data, attn.k_norm.weight.data, cache, False, pos)
    got_q = qkv[:, :nq * d].view(T, nq, d)
    got_k = qkv[:, nq * d:(nq + nk) * d].view(T, nk, d)
    assert torch.allclose(got_q, q_ref[0].transpose(0, 1), rtol=2e-5, atol=2e-6), (got_q - q_ref[0].transpose(0, 1)).abs().max()
    assert torch.allclose(got_k, k_ref[0].transpose(0, 1), rtol=2e-5, atol=2e-6)
    assert torch.equal(qkv[:, (nq + nk) * d:], v_ref)               # v untouched
