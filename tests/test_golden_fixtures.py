"""Committed golden fixtures (tests/golden/): (1) CPU -- the oracle reproduces operators_v1.npz bit for bit (the
fixtures pin the oracle against drift) and the reference's own golden vectors load; (2) GPU -- the HIP kernels, called
through the C ABI, reproduce the same fixtures (the GPU box has neither /root/reference nor a need to recompute the
expected values). Bars as everywhere: bit-exact for integer / index / cache work and for the bit-exact row-wise
operators, <= 1 ulp for the 16-bit reductions, <= 1e-3 relative L2 for attention."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
_DT = {"torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16}


def fixtures():
    z = np.load(os.path.join(HERE, "golden", "operators_v1.npz"))
    out = {}
    for k in z.files:
        if k.endswith("__dtype"):
            continue
        t = torch.from_numpy(z[k].copy())
        if k + "__dtype" in z.files:
            t = t.view(_DT[str(z[k + "__dtype"])])
        out[k] = t
    return out


def bits(t):
    return t.contiguous().view(torch.int16) if t.dtype in (torch.bfloat16, torch.float16) else t


def test_reference_goldens_file_is_well_formed():
    j = json.load(open(os.path.join(HERE, "golden", "reference_goldens.json")))
    assert len(j["qwen2_attention_prefill"]["first10"]) == 10 and len(j["qwen2_attention_decode"]["first10"]) == 10
    assert all("cite" in v for k, v in j.items() if not k.startswith("_"))


def test_oracle_reproduces_the_committed_fixtures():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    fresh, stored = mg.cases(), fixtures()
    assert set(fresh) == set(stored)
    for k in fresh:
        assert fresh[k].dtype == stored[k].dtype and torch.equal(bits(fresh[k]), bits(stored[k])), k


# ---------------------------------------------------------------------------------------------- GPU
DEV = "cuda"


def _ulp_close(got, ref, min_exact=0.99):
    got, ref = got.float().cpu(), ref.float()
    tol = 2.0 ** -8 * (ref.abs() * 2 + ref.abs().mean() * 0.5)
    assert ((got - ref).abs() <= tol).all()
    assert (got == ref).float().mean() >= min_exact


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float()
    return ((got - ref).norm() / ref.norm()).item()


@pytest.mark.gpu
def test_hip_kernels_reproduce_the_committed_fixtures():
    from xllm_amd import ops
    f = fixtures()
    d = lambda k: f[k].to(DEV)
    # row-wise
    out = torch.empty_like(d("rms_norm.x"))
    ops.rms_norm(out, d("rms_norm.x"), d("rms_norm.w"), 1e-6)
    _ulp_close(out, f["rms_norm.out"])
    x, r = d("rms_norm.x").clone(), d("fused_add_rms_norm.res_in").clone()
    ops.fused_add_rms_norm(x, r, d("rms_norm.w"), 1e-6)
    _ulp_close(x, f["fused_add_rms_norm.out"])
    assert torch.equal(bits(r.cpu()), bits(f["fused_add_rms_norm.res_out"]))
    q, k = d("rope.q").clone(), d("rope.k").clone()
    ops.rotary_embedding(d("rope.pos"), q, k, d("rope.cache"), True, head_size=64)
    assert torch.equal(bits(q.cpu()), bits(f["rope.q_out"])) and torch.equal(bits(k.cpu()), bits(f["rope.k_out"]))
    act = torch.empty(4, 128, dtype=torch.bfloat16, device=DEV)
    ops.act_and_mul(act, d("act.in"), "silu")
    _ulp_close(act, f["act.out"], min_exact=0.98)
    qa, sa = ops.scaled_quantize(d("act.out"))
    assert torch.equal(qa.cpu(), f["quant.q"]) and torch.equal(sa.cpu(), f["quant.scale"])
    # W8A8 GEMM: accumulators and epilogue
    acc = torch.empty(8, 16, dtype=torch.int32, device=DEV)
    y = ops.scaled_matmul(d("gemm.a"), d("gemm.w"), d("gemm.a_scale"), d("gemm.w_scale"), torch.bfloat16, d("gemm.bias"),
                          acc_out=acc)
    assert torch.equal(acc.cpu(), f["gemm.acc"])
    _ulp_close(y, f["gemm.out"], min_exact=0.999)
    # KV write, decode, prefill, chunked prefill
    kc, vc = d("kv.kc_in").clone(), d("kv.vc_in").clone()
    ops.reshape_paged_cache(d("kv.slots"), d("kv.k"), d("kv.v"), kc, vc)
    assert torch.equal(bits(kc.cpu()), bits(f["kv.kc_out"])) and torch.equal(bits(vc.cpu()), bits(f["kv.vc_out"]))
    scale = 1.0 / math.sqrt(64)
    od = ops.paged_attention(d("decode.q"), kc, vc, None, d("decode.kv_lens"), d("decode.block_table"), 1, 40, scale)
    assert _rel(od, f["decode.out"]) <= 1e-3
    op = ops.prefill_attention(d("prefill.q"), d("prefill.k"), d("prefill.v"), d("prefill.cu"), d("prefill.cu"), 33,
                               scale, True)
    assert _rel(op, f["prefill.out"]) <= 1e-3
    oc = ops.paged_attention(d("chunked.q"), kc, vc, d("chunked.cu_q"), d("decode.kv_lens"), d("decode.block_table"), 5,
                             40, scale, is_causal=True)
    assert _rel(oc, f["chunked.out"]) <= 1e-3
    # fp8, MoE, sampler
    qf, sf = ops.fp8_scaled_quantize(d("fp8.x"))
    assert torch.equal(qf.view(torch.uint8).cpu(), f["fp8.q"]) and torch.equal(sf.cpu(), f["fp8.scale"])
    sd, ds, sz = ops.moe_compute_index(d("moe_index.expert_id"), 8)[:3]
    assert torch.equal(sd.cpu(), f["moe_index.src_dst"]) and torch.equal(ds.cpu(), f["moe_index.dst_src"])
    assert torch.equal(sz.cpu(), f["moe_index.sizes"])
    tw, ti = ops.moe_fused_topk(d("topk.gating"), 4, True, None, "softmax")
    assert torch.equal(ti.cpu(), f["topk.ids"])
    torch.testing.assert_close(tw.cpu(), f["topk.weights"], rtol=3e-6, atol=1e-7)
    assert torch.equal(ops.philox_uniform(8, 7, 5).cpu(), f["philox.u_seed7_off5"])
    rj = ops.rejection_sample(d("rej.draft"), d("rej.n"), d("rej.cu"), d("rej.dp"), d("rej.tp"), d("rej.bonus"),
                              d("rej.ur"), d("rej.up"))
    assert torch.equal(rj.cpu(), f["rej.out"])
    # MLA over the paged latent cache: prefill (bottom-right causal) and decode of each sequence's last token
    mq, mcache, m_scale = d("mla.q"), d("mla.cache"), 192 ** -0.5
    mp = ops.mla_prefill(mq, mcache, d("mla.cu_q"), d("mla.kv_lens"), d("mla.block_table"), 512, m_scale, 130, is_causal=True)
    assert _rel(mp.view(mp.size(0), -1), f["mla.prefill_out"]) <= 1e-3
    last = f["mla.cu_q"][1:].long() - 1
    mdc = ops.mla_decode(mq[last.to(DEV)].contiguous(), mcache, d("mla.kv_lens"), d("mla.block_table"), 512, m_scale, 130)
    assert _rel(mdc.view(mdc.size(0), -1), f["mla.decode_out"]) <= 1e-3
    # grouped gate, combine, grouped GEMM
    gw, gi = ops.moe_grouped_topk(d("gtopk.gating"), 4, 4, 2, True, d("gtopk.bias"), "sigmoid", 2.5)
    assert torch.equal(gi.cpu(), f["gtopk.sig_ids"])
    torch.testing.assert_close(gw.cpu(), f["gtopk.sig_w"], rtol=3e-6, atol=1e-7)
    gw, gi = ops.moe_active_topk(d("gtopk.gating"), 3, 8, 3, False, None, "softmax", 1.0)
    assert torch.equal(gi.cpu(), f["gtopk.soft_ids"])
    torch.testing.assert_close(gw.cpu(), f["gtopk.soft_w"], rtol=3e-6, atol=1e-7)
    _ulp_close(ops.moe_combine_result(d("combine.gemm2"), d("combine.w"), 10, 2), f["combine.out"])
    _ulp_close(ops.group_gemm(d("ggemm.a"), d("ggemm.w"), d("moe_index.sizes")), f["ggemm.out"])
