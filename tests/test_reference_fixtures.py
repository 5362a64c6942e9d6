"""Pins the oracle on the fixtures the reference's own layer tests hold (tests/core/layers/mlu/*_test.cpp): hard-coded
statistics / values produced on the reference's hardware from seeded_tensor or constant inputs. The CPU tests run the ORACLE
operators; the `gpu` tests run the same fixtures through the HIP kernels (C ABI) and must give the same numbers.

 * moe_gate_test.cpp:143-272      grouped top-k gate: min / max / sum of the routing weights and of the expert ids for
                                  sigmoid + correction bias (noaux_tc), softmax, and topk_group = 1 (all-ties case)
 * dense_mlp_test.cpp:373-438     W8A8 "smoothquant" MLP on constant input: every output = 1105920.0
 * fused_moe_test.cpp:406-476     W8A8 MoE layer (16 experts, top-2 of 4 groups, one shared expert): every output = 992.0

The MLU "smooth" vectors (ScaledQuantizeParams.smooth, param.h:756-819; the DCU path passes none, linear.cpp:489-490) are an
element-wise multiply in front of the per-token quantiser; everything else is the hot path's own arithmetic: per-token int8
quantisation (RNE), exact int32 GEMM + scale epilogue, SiLU * mul with its two 16-bit roundings, the grouped gate (bias for
selection only, lowest index wins ties, renormalise, routed_scaling_factor), the weighted combine.
NOT used: mla_test.cpp:283-331 -- that layer includes the DeepSeek-V3.2 index cache / sparse indexer, an MLU-only operator set
outside SURVEY 8 (a6 is the dense MLA)."""
import pytest
import torch

from oracle import oracle as orc

BF = torch.bfloat16
GATE_CASES = [
    # name, E, H, T, topk, n_group, topk_group, route_scale, scoring, bias, (rw min, max, sum), (id min, max, sum)
    ("moe_gate_tests.sigmoid", 16, 7168, 512, 2, 4, 4, 2.5, "sigmoid", True, (1.25, 1.25, 1280.0), (6, 7, 6656)),
    ("moe_gate_tests.softmax", 16, 7168, 512, 2, 4, 4, 2.5, "softmax", False, (3.16604e-14, 2.5, 1280.0), (0, 14, 4413)),
    ("moe_gate_tests.sigmoid_topk1", 8, 1024, 128, 2, 1, 1, 1.0, "sigmoid", False, (0.5, 0.5, 128.0), (0, 1, 128)),
]


def _within(actual, expected, rtol=1e-2, atol=1e-5):   # tests_utils.cpp:90-117 expect_tensor_stats
    return abs(actual - expected) <= atol + rtol * abs(expected)


def _gate_inputs(name, E, H, T, bias):
    w = orc.seeded_tensor("moe_gate_tests.gate_proj.weight", (E, H), BF)                 # moe_gate_test.cpp:72-94
    b = orc.seeded_tensor("moe_gate_tests.e_score_correction_bias", (E,), BF) if bias else None
    x = orc.seeded_tensor(name + ".hidden_states", (T, H), BF)                          # :109-112
    return x, w, b


def _check_gate(rw, ids, rw_stats, id_stats):
    rw, ids = rw.float().cpu(), ids.cpu()
    for got, exp in zip((rw.min().item(), rw.max().item(), rw.double().sum().item()), rw_stats):
        assert _within(got, exp), (got, exp)
    # the reference allows 1 % on the id statistics too; the ids are integers: hold them exactly
    assert (ids.min().item(), ids.max().item(), ids.sum().item()) == id_stats


@pytest.mark.parametrize("case", GATE_CASES, ids=[c[0] for c in GATE_CASES])
def test_grouped_gate_reproduces_reference_fixture_stats(case):
    name, E, H, T, topk, ng, tg, scale, scoring, bias, rw_stats, id_stats = case
    x, w, b = _gate_inputs(name, E, H, T, bias)
    logits = orc.matmul(x, w)                                       # ReplicatedLinear gate_proj, bf16 (moe_gate.cpp:63-66)
    rw, ids = orc.moe_grouped_topk(logits, topk, ng, tg, True, b, scoring, scale)
    _check_gate(rw, ids, rw_stats, id_stats)


def _smooth_quant_linear(x, smooth, wq, ws, lin=None):
    xs = (x.float() * smooth).contiguous()
    if lin is None:
        q, s = orc.scaled_quantize(xs)
        return orc.scaled_matmul(q, wq, s, ws, BF)
    return lin(xs, wq, ws)


def _mlp(x, H, I, w_gate_up, w_down, lin=None, act=None):
    gu_w = torch.full((2 * I, H), w_gate_up).to(torch.int8)      # torch::full(...).to(kInt8): truncation (1.5 -> 1)
    dn_w = torch.full((H, I), w_down).to(torch.int8)
    gu = _smooth_quant_linear(x, torch.full((H,), 0.05), gu_w, torch.full((2 * I,), 0.1), lin)
    if act is None:
        a = torch.empty(x.shape[0], I, dtype=BF)
        orc.act_and_mul(a, gu, "silu")
    else:
        a = act(gu, I)
    return _smooth_quant_linear(a, torch.full((I,), 0.05), dn_w, torch.full((H,), 0.1), lin)


def test_dense_mlp_smoothquant_fixture_value():
    H, I = 7168, 9216
    y = _mlp(torch.full((16, H), 0.5).to(BF), H, I, 5.0, 3.0)
    assert torch.all(y.float() == 1105920.0), y.float().unique()


def test_fused_moe_fixture_value():
    H, I, E, T = 7168, 2048, 16, 8          # 8 of the fixture's 512 identical tokens
    x = torch.full((T, H), 0.05).to(BF)
    logits = orc.matmul(x, torch.full((E, H), 5.0).to(BF))
    rw, ids = orc.moe_grouped_topk(logits, 2, 4, 4, True, torch.full((E,), 0.1).to(BF), "sigmoid", 2.5)
    assert torch.all(rw == 1.25) and torch.equal(ids, torch.tensor([[0, 1]] * T, dtype=torch.int32))
    routed = _mlp(x, H, I, 2.0, 3.0)         # all 16 experts hold the same weights
    shared = _mlp(x, H, I, 1.5, 1.3)
    comb = orc.moe_combine(routed.repeat_interleave(2, 0).contiguous(), rw, T, 2)
    out = (comb.float() + shared.float()).to(BF)
    assert torch.all(out.float() == 992.0), out.float().unique()


# ----------------------------------------------------------------------------------------------- the same through the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("case", GATE_CASES, ids=[c[0] for c in GATE_CASES])
def test_grouped_gate_fixture_stats_through_hip(case):
    from xllm_amd import ops
    name, E, H, T, topk, ng, tg, scale, scoring, bias, rw_stats, id_stats = case
    x, w, b = _gate_inputs(name, E, H, T, bias)
    logits = ops.matmul(x.cuda(), w.cuda())
    rw, ids = ops.moe_active_topk(logits, topk, ng, tg, True, None if b is None else b.cuda(), scoring, scale)
    _check_gate(rw, ids, rw_stats, id_stats)


@pytest.mark.gpu
def test_dense_mlp_and_fused_moe_fixture_values_through_hip():
    from xllm_amd import ops

    def lin(xs, wq, ws):
        q, s = ops.scaled_quantize(xs.cuda())
        return ops.scaled_matmul(q, wq.cuda(), s, ws.cuda(), BF).cpu()

    def act(gu, I):
        a = torch.empty(gu.shape[0], I, dtype=BF, device="cuda")
        ops.act_and_mul(a, gu.cuda(), "silu")
        return a.cpu()

    H, I = 7168, 9216
    y = _mlp(torch.full((16, H), 0.5).to(BF), H, I, 5.0, 3.0, lin, act)
    assert torch.all(y.float() == 1105920.0), y.float().unique()
    H, I, T = 7168, 2048, 8
    x = torch.full((T, H), 0.05).to(BF)
    logits = ops.matmul(x.cuda(), torch.full((16, H), 5.0).to(BF).cuda())
    rw, ids = ops.moe_active_topk(logits, 2, 4, 4, True, torch.full((16,), 0.1).to(BF).cuda(), "sigmoid", 2.5)
    assert torch.all(rw == 1.25) and torch.equal(ids.cpu(), torch.tensor([[0, 1]] * T, dtype=torch.int32))
    routed, shared = _mlp(x, H, I, 2.0, 3.0, lin, act), _mlp(x, H, I, 1.5, 1.3, lin, act)
    comb = ops.moe_combine_result(routed.repeat_interleave(2, 0).contiguous().cuda(), rw, T, 2).cpu()
    out = (comb.float() + shared.float()).to(BF)
    assert torch.all(out.float() == 992.0), out.float().unique()
