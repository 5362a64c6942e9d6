"""TEST INFRASTRUCTURE ONLY (never imported by xllm_amd/): CPU restatement of the reference's routed-expert layer, composed from
the oracle's operators in the reference's order.

Follows FusedMoEImpl::forward_experts (layers/dcu/fused_moe.cpp:217-337): select_experts (gating top-k: moe_active_topk ->
cuda::moe_fused_topk for an ungrouped softmax / sigmoid gate) -> cuda::moe_compute_index -> expand (index_select of the
hidden states by dst_src / topk, :250-262) -> group_gemm w13 [E, 2 I, H] -> act_and_mul (SiLU(gate) * up, gate first) ->
group_gemm w2 [E, H, I] -> index_copy_ back to (token, k) order (:296-303) -> moe_combine_result.

Pinned (tests/test_oracle_moe.py) against an independent implementation: HuggingFace `Qwen3MoeSparseMoeBlock` (per-expert loop
with index_add_), with and without top-k renormalisation.
"""
import torch

from . import oracle as orc


def fused_moe(x, gate_w, w13, w2, topk: int, renormalize: bool, scoring_func: str = "softmax", correction_bias=None,
              num_expert_group: int = 1, topk_group: int = 1, route_scale: float = 1.0):
    """x [T, H]; gate_w [E, H]; w13 [E, 2 I, H] (gate rows first, then up); w2 [E, H, I]; returns [T, H] in x.dtype.
    num_expert_group > 1: the device-limited gate of DeepSeek-V2 / V3 (moe_active_topk -> dcu::moe_grouped_topk,
    kernels/dcu/topk_gate.cpp:59-146; fused_moe.cpp:155-166)"""
    T, H = x.shape
    E = gate_w.size(0)
    logits = orc.matmul(x, gate_w.to(x.dtype))
    if num_expert_group > 1 or route_scale != 1.0:
        weights, ids = orc.moe_grouped_topk(logits, topk, num_expert_group, topk_group, renormalize, correction_bias,
                                            scoring_func, route_scale)
    else:
        weights, ids = orc.moe_fused_topk(logits, topk, renormalize, correction_bias, scoring_func)
    src_dst, dst_src, sizes = orc.moe_compute_index(ids, E)
    xs = x.index_select(0, (dst_src // topk).long()).contiguous()          # expand, sorted by expert
    h13 = orc.group_gemm(xs, w13.to(x.dtype), sizes)
    act = torch.empty(h13.size(0), h13.size(1) // 2, dtype=x.dtype)
    orc.act_and_mul(act, h13, "silu")
    h2 = orc.group_gemm(act, w2.to(x.dtype), sizes)
    full = torch.empty_like(h2)
    full.index_copy_(0, dst_src.long(), h2)                                 # back to (token, k) order
    return orc.moe_combine(full, weights, T, topk), (weights, ids)
