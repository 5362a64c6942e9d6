"""The oracle's routed-expert layer (oracle/moe.py = FusedMoEImpl::forward_experts restated over the oracle operators) against
HuggingFace Qwen3MoeSparseMoeBlock, an independent implementation: routing weights, expert ids and the layer output, fp32."""
import pytest
import torch

from oracle import moe as omoe

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("renorm", [True, False])
@pytest.mark.parametrize("T,E,topk,H,inter", [(37, 16, 4, 64, 48), (5, 8, 2, 32, 16), (200, 128, 8, 64, 24)])
def test_routed_experts_oracle_equals_hf_qwen3_moe_block(renorm, T, E, topk, H, inter):
    from transformers.models.qwen3_moe import modeling_qwen3_moe as q
    cfg = q.Qwen3MoeConfig(hidden_size=H, moe_intermediate_size=inter, num_experts=E, num_experts_per_tok=topk,
                           norm_topk_prob=renorm, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                           intermediate_size=32, vocab_size=64)
    torch.manual_seed(T + E)
    blk = q.Qwen3MoeSparseMoeBlock(cfg).float().eval()
    for p in blk.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2)
    x = torch.randn(T, H)
    with torch.no_grad():
        ref = blk(x[None])[0]
        _, ref_w, ref_ids = blk.gate(x)
    out, (w, ids) = omoe.fused_moe(x, blk.gate.weight.detach(), blk.experts.gate_up_proj.detach(), blk.experts.down_proj.detach(),
                                   topk, renorm)
    # the same experts are selected (set equality per token: HF's topk and the oracle order ties differently at most), with the
    # same weights, and the layer output agrees to fp32 rounding
    assert torch.equal(ids.long().sort(-1).values, ref_ids.sort(-1).values)
    order_o, order_r = ids.long().argsort(-1), ref_ids.argsort(-1)
    assert torch.allclose(w.gather(-1, order_o), ref_w.float().gather(-1, order_r), rtol=1e-5, atol=1e-7)
    assert torch.allclose(out, ref, rtol=2e-4, atol=2e-5), (out - ref).abs().max()


def test_deepseek_v3_moe_layer_oracle_equals_hf():
    """DeepSeek-V3's MoE layer: grouped sigmoid gate with correction bias and routed scaling + routed experts + the shared
    experts added after (fused_moe.cpp:304-335) against HuggingFace DeepseekV3MoE"""
    from transformers.models.deepseek_v3 import modeling_deepseek_v3 as m
    H, E, topk, inter, T = 64, 32, 4, 24, 91
    cfg = m.DeepseekV3Config(hidden_size=H, n_routed_experts=E, num_experts_per_tok=topk, n_group=8, topk_group=3,
                             norm_topk_prob=True, routed_scaling_factor=2.5, n_shared_experts=1, moe_intermediate_size=inter,
                             num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, intermediate_size=32, vocab_size=64,
                             q_lora_rank=16, kv_lora_rank=16, qk_nope_head_dim=8, qk_rope_head_dim=8, v_head_dim=8)
    torch.manual_seed(5)
    moe = m.DeepseekV3MoE(cfg).float().eval()
    for p in moe.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2)
    moe.gate.e_score_correction_bias.copy_(torch.randn(E) * 0.1)
    x = torch.randn(T, H)
    with torch.no_grad():
        ref = moe(x[None])[0]
    routed, _ = omoe.fused_moe(x, moe.gate.weight.detach(), moe.experts.gate_up_proj.detach(), moe.experts.down_proj.detach(), topk,
                               True, "sigmoid", moe.gate.e_score_correction_bias.clone(), 8, 3, 2.5)
    se = moe.shared_experts
    gate_up = torch.cat([se.gate_proj.weight, se.up_proj.weight], 0).detach()          # the reference's fused gate_up layout
    from oracle import oracle as orc
    h = orc.matmul(x, gate_up)
    act = torch.empty(T, h.size(1) // 2)
    orc.act_and_mul(act, h, "silu")
    shared = orc.matmul(act, se.down_proj.weight.detach())
    out = routed + shared
    assert torch.allclose(out, ref, rtol=3e-4, atol=3e-5), (out - ref).abs().max()
