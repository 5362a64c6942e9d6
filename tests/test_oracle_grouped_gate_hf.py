"""The oracle's grouped gate (dcu::moe_grouped_topk, kernels/dcu/topk_gate.cpp:59-125: the DeepSeek-V3 device-limited gate the
reference delegates to aiter) against HuggingFace DeepseekV3TopkRouter, an independent implementation of the published
algorithm: sigmoid scores + correction bias for the choice, group score = sum of the group's top-2, topk_group groups kept,
top-k inside them, UNBIASED scores as weights, renormalised, times routed_scaling_factor. Expert sets and weights must agree."""
import pytest
import torch

from oracle import oracle as orc

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("E,n_group,topk_group,topk,renorm,scale", [(256, 8, 4, 8, True, 2.5), (64, 4, 2, 6, True, 1.0),
                                                                    (32, 8, 3, 4, False, 16.0), (16, 1, 1, 4, True, 1.0)])
def test_grouped_gate_oracle_equals_hf_deepseek_v3_router(E, n_group, topk_group, topk, renorm, scale):
    from transformers.models.deepseek_v3 import modeling_deepseek_v3 as m
    H, T = 48, 123
    cfg = m.DeepseekV3Config(hidden_size=H, n_routed_experts=E, num_experts_per_tok=topk, n_group=n_group, topk_group=topk_group,
                             norm_topk_prob=renorm, routed_scaling_factor=scale, num_hidden_layers=1, num_attention_heads=2,
                             num_key_value_heads=2, intermediate_size=32, moe_intermediate_size=16, vocab_size=64,
                             q_lora_rank=16, kv_lora_rank=16, qk_nope_head_dim=8, qk_rope_head_dim=8, v_head_dim=8)
    torch.manual_seed(E + topk)
    router = m.DeepseekV3TopkRouter(cfg).float().eval()
    torch.nn.init.normal_(router.weight, 0.0, 0.5)
    router.e_score_correction_bias.copy_(torch.randn(E) * 0.1)
    x = torch.randn(T, H)
    with torch.no_grad():
        logits, ref_w, ref_ids = router(x)
    w, ids = orc.moe_grouped_topk(logits, topk, n_group, topk_group, renorm, router.e_score_correction_bias.clone(), "sigmoid", scale)
    assert torch.equal(ids.long().sort(-1).values, ref_ids.sort(-1).values)
    assert torch.allclose(w.gather(-1, ids.long().argsort(-1)), ref_w.float().gather(-1, ref_ids.argsort(-1)), rtol=2e-6, atol=1e-7)
