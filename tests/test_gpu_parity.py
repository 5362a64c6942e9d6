"""GPU parity tests: every C-ABI kernel of the hot path against the CPU oracle on the same seeded inputs.

Bars: indexing / KV cache contents / int32 accumulators bit-exact; 16-bit row-wise outputs
within 1 ulp of the output dtype (reduction order and expf differ) and bit-identical on >= 99 % of the
elements; bf16 attention <= 1e-3 relative (L2 over the tensor) and <= 2 bf16 ulp of the row scale
element-wise; fp8 <= 2e-2.
"""
import math
import os

import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from xllm_amd import ops
DEV = "cuda"
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 2.0 ** -23}


def assert_ulp_close(got, ref, dtype, ulps=1.0, min_exact=0.99):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    # ulp of a value in [2^e, 2^(e+1)) is <= 2*eps*|x|; results that cancel towards zero get the ulp of the
    # tensor's typical magnitude as an absolute floor
    tol = ulps * ULP[dtype] * (ref.abs() * 2 + ref.abs().mean() * 0.5)
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"max err {(got - ref).abs().max()} at {bad.nonzero()[:4]}"
    exact = (got == ref).float().mean().item()
    assert exact >= min_exact, f"only {exact:.4f} bit-identical"


def rel_l2(got, ref):
    got, ref = got.double(), ref.double().to(got.device)
    return float((got - ref).norm() / ref.norm().clamp_min(1e-300))


def assert_attn_close(got, ref, rel=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).norm() / ref.norm().clamp_min(1e-30)
    assert err <= rel, f"relative L2 error {err:.3e} > {rel}"
    scale = ref.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    assert ((got - ref).abs() <= 2 * 2.0 ** -8 * scale + 1e-30).all(), f"max abs {(got - ref).abs().max()}"


def assert_prefill_close(out, q, k, v, cu, scale):
    """flash prefill against the oracle. The default kernel feeds ONE 16-bit P per score to PV, as the reference does
    (flashinfer_attention.cpp:84-90; oracle p_round); that rounding alone moves the result by e_ref ~ 2.5e-3 from the fp32-P
    result, so the bars are relative to it: no further from the exact result than the reference's own spec (x1.25), and
    within 2 e_ref of that spec (the two round P at different points: un-normalised vs normalised). With
    XLLM_MI355_PREFILL_P=2 (P = hi + lo) the fp32-P bar of 1e-3 applies."""
    import os
    ref0 = orc.attention_varlen(q, k, v, cu, cu, scale, causal=True)
    if os.environ.get("XLLM_MI355_PREFILL_P") == "2" or q.shape[-1] != 128:
        assert_attn_close(out, ref0)
        return
    ref1 = orc.attention_varlen(q, k, v, cu, cu, scale, causal=True, p_round=True)
    r = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    e_ref = r(ref1, ref0)
    assert torch.isfinite(out.float()).all()
    assert r(out, ref0) <= max(1e-3, 1.25 * e_ref), (r(out, ref0), e_ref)
    assert r(out, ref1) <= max(1e-3, 2.0 * e_ref), (r(out, ref1), e_ref)
    # round 3: ABSOLUTE bar against the oracle that rounds P where a flash kernel does (un-normalised, per 64-key tile).
    # The oracle mode restates the DEFAULT kernel's tiling; the register-staged A/B arm (XLLM_MI355_PREFILL_DMA=0) walks the keys
    # in other tiles, so its cast points differ and it stays under the two relative bars above.
    if os.environ.get("XLLM_MI355_PREFILL_DMA") == "0":
        return
    kc_, vc_ = k.contiguous(), v.contiguous()
    ref2 = orc.attention_varlen(q, kc_, vc_, cu, cu, scale, causal=True, p_round="flash")
    assert r(out, ref2) <= 1e-3, r(out, ref2)


# ------------------------------------------------------------------------------------------- probes
def test_library_loaded_is_hip_path():
    """the library the GPU tests call is the HIP one, at the ABI version the bindings were written for, AND built from the
    sources that were shipped next to it (round-4 review, weak #10: the driver's GPU run uses the builder's prebuilt .so; the
    digest of xllm_amd/csrc + include/xllm_mi355.h is compiled into it)"""
    import importlib.util
    from xllm_amd import _lib
    assert _lib.lib().xllm_mi355_abi_version() == _lib.ABI_VERSION == 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("source_digest", os.path.join(root, "tools", "source_digest.py"))
    sd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sd)
    assert _lib.lib().xllm_mi355_build_digest().decode() == sd.lib_digest(), "libxllm_mi355.so was not built from these sources"


# ------------------------------------------------------------------------------------------- MLA weight absorption
@pytest.mark.parametrize("T,H,K,N,dtype,strided", [(128, 16, 128, 512, torch.bfloat16, True),    # cfg4 rank: q_nope x W_kc
                                                    (128, 16, 512, 128, torch.bfloat16, False),   # cfg4 rank: attn x W_vc (project_output)
                                                    (77, 3, 64, 40, torch.float16, True),         # ragged tokens, columns not a tile multiple
                                                    (1, 128, 128, 512, torch.bfloat16, False),    # one token, every head of the model
                                                    (300, 2, 512, 128, torch.bfloat16, True)])
def test_bmm_heads_matches_the_oracle(T, H, K, N, dtype, strided):
    """torch::bmm(x.transpose(0, 1), w).transpose(0, 1) of DeepseekV2AttentionImpl (deepseek_v2_attention.cpp:180-187, 310-311) on
    this backend's per-head GEMM: against the oracle's F::linear per head (fp32 sequential sums, one rounding). x is a slice of a
    wider packed tensor and w a strided view over heads when `strided` (what the layer passes: q[..., :nope], kv_b_proj's slice)."""
    g = torch.Generator().manual_seed(T * 131 + N)
    xw = torch.randn(T, H, K + 64, generator=g).to(dtype)
    x = xw[..., :K] if strided else xw[..., :K].contiguous()
    ww = (torch.randn(H, N + 16, K, generator=g) / math.sqrt(K)).to(dtype)
    w = ww[:, 8:8 + N] if strided else ww[:, 8:8 + N].contiguous()
    ref = orc.bmm_heads(x, w)
    xd, wd = xw.to(DEV)[..., :K], ww.to(DEV)[:, 8:8 + N]
    if not strided:
        xd, wd = xd.contiguous(), wd.contiguous()
    got = ops.bmm_heads(xd, wd)
    assert got.shape == (T, H, N)
    assert_ulp_close(got, ref, dtype, ulps=1.0, min_exact=0.98)
    # into a strided output (the layer writes the first kv_lora columns of the attention kernel's [T, h, 576] input)
    buf = torch.full((T, H, N + 64), 7.0, dtype=dtype, device=DEV)
    ops.bmm_heads(xd, wd, out=buf[..., :N])
    assert torch.equal(buf[..., :N], got) and bool((buf[..., N:] == 7.0).all())
    # the vendor-library form it replaces agrees to the last bit or so (same products, another summation order)
    vend = torch.bmm(xd.transpose(0, 1), wd.transpose(1, 2)).transpose(0, 1)
    assert_ulp_close(got, vend.cpu(), dtype, ulps=2.0, min_exact=0.9)


def test_bmm_heads_declines_what_it_cannot_address():
    from xllm_amd._lib import Mi355Error
    x = torch.randn(8, 2, 48, device=DEV).bfloat16()          # K % 32 != 0
    w = torch.randn(2, 16, 48, device=DEV).bfloat16()
    with pytest.raises(Mi355Error):
        ops.bmm_heads(x, w)
    x = torch.randn(8, 2, 68, device=DEV).bfloat16()[..., 4:]  # rows start off a 16-byte boundary
    w = torch.randn(2, 16, 64, device=DEV).bfloat16()
    with pytest.raises(Mi355Error):
        ops.bmm_heads(x, w)


# ------------------------------------------------------------------------------------------- KV block copy
@pytest.mark.parametrize("shape,dtype,with_v", [((128, 4, 128), torch.bfloat16, True),     # Qwen2-7B page: 128 KiB per block
                                                ((16, 2, 64), torch.float16, True),
                                                ((64, 1, 576), torch.bfloat16, False),     # MLA latent cache, K only
                                                ((3, 1, 5), torch.float32, True),          # 60 bytes: 4-byte units
                                                ((3, 1, 5), torch.bfloat16, True),         # 30 bytes: 2-byte units
                                                ((5, 1, 3), torch.int8, True)])            # 15 bytes: byte units
def test_block_copy_bit_exact(shape, dtype, with_v):
    """cuda::block_copy (worker_impl.cpp:1071-1082): fan-out copies of whole cache blocks over every layer, against the oracle"""
    L, nb = 5, 40
    g = torch.Generator().manual_seed(99)
    mk = lambda: [(torch.randn(nb, *shape, generator=g) * 20).to(dtype) for _ in range(L)]
    k, v = mk(), (mk() if with_v else None)
    perm = torch.randperm(nb, generator=g)
    src = perm[:6].to(torch.int32)
    fan = [1, 3, 1, 7, 2, 4]
    dst = perm[6:6 + sum(fan)].to(torch.int32)
    cs = torch.tensor(fan).cumsum(0).to(torch.int32)
    kd = [t.to(DEV) for t in k]
    vd = [t.to(DEV) for t in v] if with_v else None
    orc.block_copy(k, v, src, dst, cs)
    kp = torch.tensor([t.data_ptr() for t in kd], dtype=torch.int64, device=DEV)
    vp = torch.tensor([t.data_ptr() for t in vd], dtype=torch.int64, device=DEV) if with_v else None
    ops.block_copy(kp, vp, src.to(DEV), dst.to(DEV), cs.to(DEV), kd[0][0].numel(), dtype)
    torch.cuda.synchronize()
    for l in range(L):
        assert torch.equal(kd[l].cpu().view(torch.uint8), k[l].view(torch.uint8))
        if with_v:
            assert torch.equal(vd[l].cpu().view(torch.uint8), v[l].view(torch.uint8))
    # nothing to do is not an error
    ops.block_copy(kp, vp, src[:0].to(DEV), dst[:0].to(DEV), cs[:0].to(DEV), kd[0][0].numel(), dtype)


def test_block_copy_at_cache_bases_that_are_not_16_byte_aligned():
    """round-5 advisor: the unit width comes from the block size alone, the cache base addresses from device arrays; a cache that is
    a VIEW at an 8-byte (2-byte) offset must still copy bit-exactly (byte-wise walk of the same units), not fault"""
    L, nb, shape, dtype = 3, 12, (16, 2, 64), torch.bfloat16          # 4096 bytes per block: the 16-byte unit is selected
    g = torch.Generator().manual_seed(5)
    numel = nb * 16 * 2 * 64
    for off in (4, 1):                                                # element offsets: 8 and 2 bytes
        backing = [(torch.randn(numel + 8, generator=g) * 9).to(dtype) for _ in range(2 * L)]
        dev = [t.to(DEV) for t in backing]
        k = [t[off:off + numel].view(nb, *shape).clone() for t in backing[:L]]
        v = [t[off:off + numel].view(nb, *shape).clone() for t in backing[L:]]
        kd = [t[off:off + numel].view(nb, *shape) for t in dev[:L]]
        vd = [t[off:off + numel].view(nb, *shape) for t in dev[L:]]
        assert kd[0].data_ptr() % 16 != 0
        src = torch.tensor([1, 7], dtype=torch.int32)
        dst = torch.tensor([0, 3, 9], dtype=torch.int32)
        cs = torch.tensor([2, 3], dtype=torch.int32)
        orc.block_copy(k, v, src, dst, cs)
        kp = torch.tensor([t.data_ptr() for t in kd], dtype=torch.int64, device=DEV)
        vp = torch.tensor([t.data_ptr() for t in vd], dtype=torch.int64, device=DEV)
        ops.block_copy(kp, vp, src.to(DEV), dst.to(DEV), cs.to(DEV), kd[0][0].numel(), dtype)
        torch.cuda.synchronize()
        for l in range(L):
            assert torch.equal(kd[l].cpu().view(torch.uint8), k[l].view(torch.uint8))
            assert torch.equal(vd[l].cpu().view(torch.uint8), v[l].view(torch.uint8))
            # nothing outside the view was touched
            assert torch.equal(dev[l][:off].cpu(), backing[l][:off]) and torch.equal(dev[l][off + numel:].cpu(), backing[l][off + numel:])


# ------------------------------------------------------------------------------------------- KV write
@pytest.mark.parametrize("nkv,d,bs,dtype", [(4, 128, 128, torch.bfloat16), (8, 128, 16, torch.bfloat16),
                                            (2, 64, 16, torch.float16), (1, 128, 64, torch.bfloat16),
                                            (4, 96, 16, torch.float32)])
def test_reshape_paged_cache_bit_exact(nkv, d, bs, dtype):
    T, nb = 77, 20
    g = torch.Generator().manual_seed(2026)
    qkv = torch.randn(T, (5 + 2 * nkv) * d, generator=g).to(dtype)
    k = qkv[:, 5 * d:(5 + nkv) * d].unflatten(-1, (nkv, d))
    v = qkv[:, (5 + nkv) * d:].unflatten(-1, (nkv, d))
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int32)
    slots[3] = -1
    kc = torch.randn(nb, bs, nkv, d, generator=g).to(dtype)
    vc = torch.randn(nb, bs, nkv, d, generator=g).to(dtype)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    orc.reshape_paged_cache(slots, k, v, kc_ref, vc_ref)
    qkv_d = qkv.to(DEV)
    kd = qkv_d[:, 5 * d:(5 + nkv) * d].unflatten(-1, (nkv, d))
    vd = qkv_d[:, (5 + nkv) * d:].unflatten(-1, (nkv, d))
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    ops.reshape_paged_cache(slots.to(DEV), kd, vd, kc_d, vc_d)
    assert torch.equal(kc_d.cpu(), kc_ref) and torch.equal(vc_d.cpu(), vc_ref)


def test_decode_metadata_update_bit_exact():
    """N2: device-side metadata refresh == the oracle on the same staging buffers, incl. the dense block table"""
    from tests.test_oracle_ops import _decode_metadata_case
    for B, Bp in [(5, 8), (256, 256), (1, 4)]:
        src, dst, seq_lens, blocks, md = _decode_metadata_case(seed=B, B=B, B_padded=Bp)
        n_idx = md["paged_kv_indices"].numel()
        dst_dev = {k: v.clone().to(DEV) for k, v in dst.items()}
        orc.decode_metadata_update(src, dst, B, Bp, B, n_idx, Bp)
        ops.decode_metadata_update({k: v.to(DEV) for k, v in src.items()}, dst_dev, B, Bp, B, n_idx, Bp)
        for k in dst:
            assert torch.equal(dst_dev[k].cpu(), dst[k]), (B, k)


def test_build_block_table_bit_exact():
    md = orc.build_batch_metadata([33, 16, 1, 40], [1, 16, 1, 8], [[5, 0, 9], [7], [3], [2, 11, 4]], 16)
    ref = orc.build_block_table_from_paged_kv(md["paged_kv_indptr"], md["paged_kv_indices"])
    got = ops.build_block_table_from_paged_kv(md["paged_kv_indptr"].to(DEV), md["paged_kv_indices"].to(DEV))
    assert torch.equal(got.cpu(), ref)


# ------------------------------------------------------------------------------------------- row-wise
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("T,H", [(1, 128), (19, 896), (64, 3584), (3, 1000), (2, 8200)])
def test_rms_norm(dtype, T, H):
    g = torch.Generator().manual_seed(T * H)
    x = torch.randn(T, H, generator=g).to(dtype)
    w = (torch.rand(H, generator=g) + 0.5).to(dtype)
    ref = torch.empty_like(x)
    orc.rms_norm(ref, x, w, 1e-6)
    out = torch.empty(T, H, dtype=dtype, device=DEV)
    ops.rms_norm(out, x.to(DEV), w.to(DEV), 1e-6)
    # fp32: the reference's own bar is 1e-5 (dcu/norm_test.cpp:54-65); inv differs by reduction order
    assert_ulp_close(out, ref, dtype, ulps=1.0 if dtype != torch.float32 else 8.0,
                     min_exact=0.98 if dtype != torch.float32 else 0.3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H", [(7, 3584), (2, 1000)])
def test_fused_add_rms_norm(dtype, T, H):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, H, generator=g).to(dtype)
    r = torch.randn(T, H, generator=g).to(dtype)
    w = (torch.rand(H, generator=g) + 0.5).to(dtype)
    xr, rr = x.clone(), r.clone()
    orc.fused_add_rms_norm(xr, rr, w, 1e-6)
    xd, rd = x.to(DEV), r.to(DEV)
    ops.fused_add_rms_norm(xd, rd, w.to(DEV), 1e-6)
    assert torch.equal(rd.cpu(), rr)  # 16-bit add is exact
    assert_ulp_close(xd, xr, dtype, min_exact=0.98)


def test_rms_norm_fp8_quant_variants():
    T, H = 9, 3584
    g = torch.Generator().manual_seed(11)
    x = torch.randn(T, H, generator=g).bfloat16()
    r = torch.randn(T, H, generator=g).bfloat16()
    w = (torch.rand(H, generator=g) + 0.5).bfloat16()
    scale = torch.tensor([0.015])
    ref = torch.empty(T, H, dtype=torch.uint8)
    orc.rms_norm_static_fp8_quant(ref, x, w, scale, 1e-6)
    out = torch.empty(T, H, dtype=torch.uint8, device=DEV)
    ops.rms_norm_static_fp8_quant(out, x.to(DEV), w.to(DEV), scale.to(DEV), 1e-6)
    d = (orc.e4m3_to_f32(out.cpu()) - orc.e4m3_to_f32(ref)).abs()
    assert (out.cpu() != ref).float().mean() < 5e-3 and d.max() <= 32  # rare 1-step flips from inv rounding
    rr = r.clone()
    ref2 = torch.empty(T, H, dtype=torch.uint8)
    orc.rms_norm_static_fp8_quant(ref2, x, w, scale, 1e-6, residual=rr)
    rd = r.to(DEV)
    out2 = torch.empty(T, H, dtype=torch.uint8, device=DEV)
    ops.fused_add_rms_norm_static_fp8_quant(out2, x.to(DEV), rd, w.to(DEV), scale.to(DEV), 1e-6)
    assert torch.equal(rd.cpu(), rr)
    assert (out2.cpu() != ref2).float().mean() < 5e-3


def test_rms_norm_dynamic_int8_fusion_equals_two_ops():
    T, H = 12, 3584
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, H, generator=g).bfloat16().to(DEV)
    r = torch.randn(T, H, generator=g).bfloat16().to(DEV)
    w = (torch.rand(H, generator=g) + 0.5).bfloat16().to(DEV)
    x2, r2 = x.clone(), r.clone()
    ops.fused_add_rms_norm(x2, r2, w, 1e-6)
    q_ref, s_ref = ops.scaled_quantize(x2)
    r3 = r.clone()
    q, s = ops.rms_norm_dynamic_int8_quant(x, w, 1e-6, residual=r3)
    assert torch.equal(r3, r2) and torch.equal(q, q_ref) and torch.equal(s, s_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("neox", [True, False])
def test_rope_bit_exact(dtype, neox):
    T, nq, nk, d = 33, 28, 4, 128
    g = torch.Generator().manual_seed(3)
    cache = orc.build_cos_sin_cache(512, d, 1000000.0, dtype)
    qkv = torch.randn(T, (nq + 2 * nk) * d, generator=g).to(dtype)
    pos = torch.randint(0, 512, (T,), generator=g)
    ref = qkv.clone()
    orc.rotary_embedding(pos, ref[:, :nq * d], ref[:, nq * d:(nq + nk) * d], cache, d, is_neox=neox)
    dev = qkv.to(DEV)
    ops.rotary_embedding(pos.to(DEV), dev[:, :nq * d], dev[:, nq * d:(nq + nk) * d], cache.to(DEV), neox, head_size=d)
    assert torch.equal(dev.cpu().view(torch.int16 if dtype != torch.float32 else torch.int32),
                       ref.view(torch.int16 if dtype != torch.float32 else torch.int32))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("mode", ["silu", "gelu", "gelu_tanh"])
def test_act_and_mul(dtype, mode):
    T, d = 5, 18944 if mode == "silu" else 1003
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(T, 2 * d, generator=g) * 2).to(dtype)
    ref = torch.empty(T, d, dtype=dtype)
    orc.act_and_mul(ref, x, mode)
    out = torch.empty(T, d, dtype=dtype, device=DEV)
    ops.act_and_mul(out, x.to(DEV), mode)
    if dtype == torch.float32:  # device expf/erff/tanhf vs glibc: reference bar 1e-5/1e-6 (dcu/activation_test.cpp:84-85)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)
    else:
        assert_ulp_close(out, ref, dtype, ulps=2.0, min_exact=0.98)


def test_fused_qk_norm_rope():
    T, nq, nk, d = 21, 8, 2, 128
    g = torch.Generator().manual_seed(13)
    qkv = torch.randn(T, (nq + 2 * nk) * d, generator=g).bfloat16()
    qw = (torch.rand(d, generator=g) + 0.5).bfloat16()
    kw = (torch.rand(d, generator=g) + 0.5).bfloat16()
    cache = orc.build_cos_sin_cache(256, d, 10000.0, torch.float32)
    pos = torch.randint(0, 256, (T,), generator=g)
    for inter in (False, True):
        ref = qkv.clone()
        orc.fused_qk_norm_rope(ref, nq, nk, nk, d, 1e-6, qw, kw, cache, inter, pos)
        dev = qkv.to(DEV)
        ops.fused_qk_norm_rope(dev, nq, nk, nk, d, 1e-6, qw.to(DEV), kw.to(DEV), cache.to(DEV), inter, pos.to(DEV))
        assert torch.equal(dev[:, (nq + nk) * d:].cpu(), qkv[:, (nq + nk) * d:])  # v untouched
        # reference tolerance 2e-3 (neox) / 2e-2 (interleaved): dcu/fused_qknorm_rope_test.cpp:172,227
        torch.testing.assert_close(dev.float().cpu(), ref.float(), rtol=2e-2, atol=2e-2)
        assert (dev.cpu() != ref).float().mean() < 0.02


# ------------------------------------------------------------------------------------------- int8 path
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", [(256, 3584), (3, 18944), (5, 120)])
def test_scaled_quantize_bit_exact(dtype, M, K):
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 3).to(dtype)
    x[0] = 0
    q_ref, s_ref = orc.scaled_quantize(x)
    q, s = ops.scaled_quantize(x.to(DEV))
    assert torch.equal(q.cpu(), q_ref) and torch.equal(s.cpu(), s_ref)


@pytest.mark.parametrize("M,N,K", [(256, 512, 3584), (7, 130, 256), (129, 257, 1024), (256, 4608, 3584), (16, 3584, 18944),
                                   (640, 384, 1024), (1030, 130, 512), (300, 3584, 3584)])
def test_scaled_matmul_int32_exact_and_epilogue(M, N, K):
    g = torch.Generator().manual_seed(M * N)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    a_s = torch.rand(M, generator=g) * 0.05 + 0.01
    w_s = torch.rand(N, generator=g) * 0.02 + 0.01
    bias = torch.randn(N, generator=g).bfloat16()
    acc_d = torch.empty(M, N, dtype=torch.int32, device=DEV)
    out = ops.scaled_matmul(a.to(DEV), w.to(DEV), a_s.to(DEV), w_s.to(DEV), torch.bfloat16, bias.to(DEV),
                            acc_out=acc_d)
    # exact reference via fp64-safe chunks (|acc| < 2^31)
    acc_ref = (a.to(DEV).double() @ w.to(DEV).double().T).to(torch.int32)
    assert torch.equal(acc_d, acc_ref)
    if M * N * K <= 256 * 512 * 3584:
        ref, acc_o = orc.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias, want_acc=True)
        assert torch.equal(acc_d.cpu(), acc_o)
        assert_ulp_close(out, ref, torch.bfloat16, min_exact=0.999)
    # split-K path (workspace registered by ops): same bits as the single-pass epilogue
    out2 = ops.scaled_matmul(a.to(DEV), w.to(DEV), a_s.to(DEV), w_s.to(DEV), torch.bfloat16, bias.to(DEV))
    assert torch.equal(out2, out)


@pytest.mark.parametrize("M,N,K,dt", [(2048, 3584, 3584, torch.bfloat16), (8192, 3584, 1024, torch.bfloat16),
                                        (2049, 520, 512, torch.float16), (256, 3584, 3584, torch.bfloat16),
                                        (1030, 130, 512, torch.bfloat16), (64, 512, 18944, torch.float16)])
def test_scaled_matmul_with_addend_is_matmul_then_16bit_add(M, N, K, dt):
    """ScaledMatmulParams::c with alpha = beta = 1 (kernels/param.h:852-866): out = r16(r16(a @ b ...) + c). The prefill shapes take
    the addend in the 8-phase kernel's dequant epilogue (xllm_mi355_scaled_matmul_add), the others run the product and one add pass:
    both bit-identical to the oracle's scaled_matmul followed by a 16-bit add, in place (output = c) and out of place."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.05 + 0.01).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.01).to(DEV)
    bias = torch.randn(N, generator=g).to(dt).to(DEV)
    c = (torch.randn(M, N, generator=g) * 3).to(dt).to(DEV)
    y = ops.scaled_matmul(a, w, a_s, w_s, dt, bias)                      # the operator without c (parity-tested above)
    want = (y.float() + c.float()).to(dt)                                # torch's 16-bit add: f32 sum, one rounding
    if M * N * K <= 256 * 3584 * 3584:                                   # ... and the oracle end to end where it finishes quickly
        ref = orc.scaled_matmul(a.cpu(), w.cpu(), a_s.cpu(), w_s.cpu(), dt, bias.cpu())
        assert torch.equal((ref.float() + c.cpu().float()).to(dt), want.cpu())
    out = ops.scaled_matmul(a, w, a_s, w_s, dt, bias, c=c)
    assert torch.equal(out, want)
    assert torch.equal(ops.add_(y, c), want)
    c_in_place = c.clone()
    got = ops.scaled_matmul(a, w, a_s, w_s, dt, bias, output=c_in_place, c=c_in_place)
    assert got.data_ptr() == c_in_place.data_ptr() and torch.equal(c_in_place, want)
    with pytest.raises(ops.Mi355Error):
        ops.scaled_matmul(a, w, a_s, w_s, dt, bias, c=c, beta=0.5)


def test_splitk_workspace_invariant_across_shapes():
    """split-K paths (prefill-sized and decode-sized) share one workspace that must be all-zero between calls"""
    g = torch.Generator().manual_seed(4)
    shapes = [(640, 384, 1024), (256, 3584, 3584), (16, 512, 18944), (1030, 130, 512), (256, 4608, 3584)]
    for rep in range(2):
        for M, N, K in shapes:
            a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
            w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
            a_s, w_s = torch.rand(M, generator=g).to(DEV), torch.rand(N, generator=g).to(DEV)
            acc = torch.empty(M, N, dtype=torch.int32, device=DEV)
            ref = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, acc_out=acc)  # single pass, no split-K
            out = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)               # planner may split K
            assert torch.equal(out, ref), (rep, M, N, K)
            assert torch.equal(acc, (a.double() @ w.double().T).to(torch.int32))


def test_w8a8_dynamic_linear_vs_dequantised_matmul():
    # tests/core/layers/npu_torch/linear_w8a8_dynamic_tests.cpp:110,125-137 (5e-2)
    M, N, K = 64, 512, 3584
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = orc.seeded_tensor("w8a8.weight", (N, K), torch.int8).to(DEV)
    w_s = (orc.seeded_tensor("w8a8.scale", (N,), torch.float32) * 0.02 + 0.01).to(DEV)
    q, s = ops.scaled_quantize(x)
    y = ops.scaled_matmul(q, w, s, w_s, torch.bfloat16)
    ref = x.float() @ (w.float() * w_s[:, None]).T
    assert ((y.float() - ref).abs() / (ref.abs() + ref.abs().mean())).max() < 5e-2


# ------------------------------------------------------------------------------------------- fp8 path
def test_fp8_quant_static_and_dynamic():
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(37, 3584, generator=g) * 4).bfloat16()
    q_ref, s_ref = orc.fp8_scaled_quantize(x)
    q, s = ops.fp8_scaled_quantize(x.to(DEV))
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.view(torch.uint8).cpu(), q_ref)
    scale = torch.tensor([0.05])
    out = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    ops.static_scaled_fp8_quant(out, x.to(DEV), scale.to(DEV))
    assert torch.equal(out.cpu(), orc.static_scaled_fp8_quant(x, scale))
    # the workspace-free entry (memset + atomic amax + scale + quantise: four graph nodes) and the two-launch form ops uses
    # (xllm_mi355_fp8_scaled_quantize_ws, round 6) give the same bits; odd sizes (scalar tail), f16, a tensor of zeros
    from xllm_amd import _lib
    for shape, dt_ in (((37, 3584), torch.bfloat16), ((3, 1001), torch.float16), ((128, 1536), torch.bfloat16), ((5, 64), torch.bfloat16)):
        for zero in (False, True):
            xd = torch.zeros(shape, dtype=dt_, device=DEV) if zero else (torch.randn(shape, generator=g) * 3).to(dt_).to(DEV)
            q2, s2 = ops.fp8_scaled_quantize(xd)
            q1 = torch.empty(shape, dtype=torch.uint8, device=DEV)
            s1 = torch.empty(1, dtype=torch.float32, device=DEV)
            rc = _lib.lib().xllm_mi355_fp8_scaled_quantize(q1.data_ptr(), xd.data_ptr(), 0, s1.data_ptr(), xd.numel(),
                                                           ops._dt(xd), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            assert torch.equal(s1, s2) and torch.equal(q1, q2.view(torch.uint8))


@pytest.mark.parametrize("per_token,per_channel", [(False, False), (True, True)])
def test_fp8_scaled_matmul(per_token, per_channel):
    M, N, K = 130, 260, 512
    g = torch.Generator().manual_seed(23)
    a = (torch.randn(M, K, generator=g) * 2).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    a_s = (torch.rand(M if per_token else 1, generator=g) * 0.05 + 0.01)
    w_s = (torch.rand(N if per_channel else 1, generator=g) * 0.02 + 0.01)
    bias = torch.randn(N, generator=g).bfloat16()
    ref = orc.fp8_scaled_matmul(a.view(torch.uint8), w.view(torch.uint8), a_s, w_s, torch.bfloat16, bias)
    out = ops.fp8_scaled_matmul(a.to(DEV), w.to(DEV), a_s.to(DEV), w_s.to(DEV), torch.bfloat16, bias.to(DEV))
    # BASELINE bar for fp8: <= 2e-2 relative; accumulation order only => far tighter in practice
    assert_ulp_close(out, ref, torch.bfloat16, ulps=1.0, min_exact=0.97)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_matmul_16bit(dtype):
    M, N, K = 70, 300, 896
    g = torch.Generator().manual_seed(29)
    a = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g).to(dtype)
    ref = orc.matmul(a, w, b)
    out = ops.matmul(a.to(DEV), w.to(DEV), b.to(DEV))
    assert_ulp_close(out, ref, dtype, ulps=2.0, min_exact=0.95)  # MFMA block-sum order vs serial fp32 sum


@pytest.mark.parametrize("M", [1, 15, 16, 17, 32, 33, 48, 64])
@pytest.mark.parametrize("N,K,dtype", [(256, 512, torch.bfloat16), (4608, 3584, torch.bfloat16), (3584, 2048, torch.float16)])
def test_matmul_weight_stream_rows(M, N, K, dtype):
    """the weight-stream kernel of the 16-bit decode linears (gemm_wsb.hip: M <= 64, N % 64 == 0): every row-block count,
    K slices through fp32 slabs (few columns) and none (many), bias, against the oracle (fp32 sum in another order) and the
    fp64 product; identical bits run to run"""
    g = torch.Generator().manual_seed(M * 31 + N)
    a = (torch.randn(M, K, generator=g) * 0.7).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g).to(dtype)
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    out = ops.matmul(ad, wd, bd)
    assert torch.equal(out, ops.matmul(ad, wd, bd))
    assert_ulp_close(out, orc.matmul(a, w, b), dtype, ulps=1.0, min_exact=0.97)
    ref64 = (a.double() @ w.double().T + b.double())
    assert ((out.cpu().double() - ref64).abs() <= ULP[dtype] * (ref64.abs() * 2 + 0.5 * ref64.abs().mean())).all()
    assert torch.equal(ops.matmul(ad, wd), ops.matmul(ad, wd, torch.zeros_like(bd)))      # no bias == zero bias


@pytest.mark.parametrize("H,N", [(512, 384), (256, 512), (512, 768), (1024, 512)])
def test_group_gemm_few_rows_per_expert(H, N):
    """MoE decode shape: a handful of rows per expert (gemm_wsb.hip grouped form): experts without a row, experts with more
    than one 16-row pass, the expand fused as an index (group_gemm_gather == index_select + group_gemm bit for bit); K = H <= 512
    with N % 256 == 0 takes the per-wave column tiles (round 6: w2 of a decode step), the others the K split over the waves"""
    T, topk, E = 24, 4, 32
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(0, E, (T, topk), generator=g, dtype=torch.int32)
    ids[:20, 0] = 5                                           # expert 5 gets >= 20 rows: two passes of 16
    ids[ids == 7] = 8                                         # expert 7 gets none
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.to(DEV), E)
    assert int(sizes[5]) >= 20 and int(sizes[7]) == 0
    x = torch.randn(T, H, generator=g).bfloat16()
    w = (torch.randn(E, N, H, generator=g) / math.sqrt(H)).bfloat16()
    xs = x[(dst_src.cpu().long() // topk)]
    got = ops.group_gemm(xs.to(DEV), w.to(DEV), sizes)
    ref = orc.group_gemm(xs, w, sizes.cpu())
    assert_ulp_close(got, ref, torch.bfloat16, ulps=1.0, min_exact=0.97)
    fused = ops.group_gemm_gather(x.to(DEV), dst_src, topk, w.to(DEV), sizes)
    if os.environ.get("XLLM_MI355_WSB", "1") != "0":          # (the tiled fallback declines the fused expand at this size)
        assert fused is not None
    assert fused is None or torch.equal(fused, got)


# ------------------------------------------------------------------------------------------- attention
def _paged_case(B, nq, nkv, d, bs, kv_lens, q_lens, dtype, seed, noise=1.0):
    g = torch.Generator().manual_seed(seed)
    pages = [(L + bs - 1) // bs for L in kv_lens]
    nb = sum(pages) + 3
    perm = torch.randperm(nb, generator=g).tolist()  # non-contiguous, shuffled pages (block id 0 is valid)
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, q_lens, blocks, bs)
    kc = (torch.randn(nb, bs, nkv, d, generator=g) * noise).to(dtype)
    vc = (torch.randn(nb, bs, nkv, d, generator=g) * noise).to(dtype)
    q = torch.randn(sum(q_lens), nq, d, generator=g).to(dtype)
    return md, kc, vc, q


DECODE_CASES = [
    # B, nq, nkv, d, bs, kv_lens
    (4, 28, 4, 128, 128, [257, 128, 1, 700]),          # Qwen2-7B TP=1 heads, ragged incl. len 1
    (3, 14, 2, 128, 128, [513, 31, 1200]),             # TP=2
    (5, 7, 1, 128, 128, [4096, 100, 129, 33, 2048]),   # TP=4: one kv head, 4 sub-ranges per workgroup
    (4, 16, 8, 128, 16, [257, 257, 16, 45]),           # MLU golden geometry (block 16 => per-lane page gather)
    (2, 14, 2, 64, 16, [130, 77]),                     # Qwen2-0.5B heads (d=64)
    (2, 32, 4, 128, 64, [300, 64]),                    # group of 8
    (1, 16, 1, 128, 128, [1000]),                      # full 16-wide group (MLA-like stacking)
]


@pytest.mark.parametrize("case", DECODE_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_paged_decode_attention(case, dtype):
    B, nq, nkv, d, bs, kv_lens = case
    md, kc, vc, q = _paged_case(B, nq, nkv, d, bs, kv_lens, [1] * B, dtype, seed=sum(kv_lens))
    scale = d ** -0.5
    ref = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale)
    out = ops.paged_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), None, md["kv_seq_lens"].to(DEV),
                              md["block_tables"].to(DEV), 1, max(kv_lens), scale)
    if dtype == torch.bfloat16:
        assert_attn_close(out, ref)
    else:
        assert_attn_close(out, ref, rel=2e-4)


def test_paged_decode_split_kv_and_garbage_tail(monkeypatch):
    """forces several grid-level splits and poisons the unused tail of the last page with NaN/Inf."""
    B, nq, nkv, d, bs = 3, 28, 4, 128, 128
    kv_lens = [1000, 130, 2500]
    md, kc, vc, q = _paged_case(B, nq, nkv, d, bs, kv_lens, [1] * B, torch.bfloat16, seed=77)
    for b, L in enumerate(kv_lens):
        last = int(md["block_tables"][b, (L - 1) // bs])
        if L % bs:
            kc[last, L % bs:] = float("nan")
            vc[last, L % bs:] = float("inf")
    scale = d ** -0.5
    ref = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale)
    out = ops.paged_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), None, md["kv_seq_lens"].to(DEV),
                              md["block_tables"].to(DEV), 1, max(kv_lens), scale)
    assert_attn_close(out, ref)


def test_paged_decode_window():
    B, nq, nkv, d, bs = 2, 8, 2, 128, 128
    kv_lens = [600, 90]
    md, kc, vc, q = _paged_case(B, nq, nkv, d, bs, kv_lens, [1] * B, torch.bfloat16, seed=5)
    scale = d ** -0.5
    ref = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                              window_left=100)
    out = ops.paged_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), None, md["kv_seq_lens"].to(DEV),
                              md["block_tables"].to(DEV), 1, max(kv_lens), scale, window_left=100)
    assert_attn_close(out, ref)


def test_softmax_rescale_branch_forced():
    """a late key with a huge score forces the running max to jump mid-stream (online-softmax rescale)."""
    B, nq, nkv, d, bs = 1, 4, 1, 128, 128
    kv_lens = [640]
    md, kc, vc, q = _paged_case(B, nq, nkv, d, bs, kv_lens, [1], torch.bfloat16, seed=3, noise=0.1)
    page = int(md["block_tables"][0, 3])
    kc[page, 17, 0] = (q[0, 2] * 4).to(torch.bfloat16)  # spike for head 2 at token 3*128+17
    scale = d ** -0.5
    ref = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale)
    out = ops.paged_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), None, md["kv_seq_lens"].to(DEV),
                              md["block_tables"].to(DEV), 1, 640, scale)
    assert_attn_close(out, ref)


PREFILL_CASES = [
    (28, 4, 128, [128, 5, 300]),
    (16, 8, 128, [128, 128]),
    (14, 2, 64, [77, 200]),
    (4, 4, 128, [513]),
    # round 6, the block -> XCD mapping of the head-dim-128 kernels (pf_block_coords): P = sequences x kv heads pairs dealt over 8 XCDs
    (8, 1, 128, [200, 130, 64]),      # P = 3: no even deal, the plain order
    (6, 2, 128, [150, 90, 300]),      # P = 6: the same
    (7, 1, 128, [260]),               # P = 1: the pair's blocks dealt over all 8 XCDs
    (8, 2, 128, [100]),               # P = 2: over 4 XCDs each
    (12, 4, 128, [40, 70]),           # P = 8: one pair per XCD
    (4, 2, 128, [33, 65, 129, 257, 64]),   # P = 10: two pairs on XCDs 0 and 1, one elsewhere
]


@pytest.mark.parametrize("nq,nkv,d,lens", PREFILL_CASES)
def test_prefill_attention(nq, nkv, d, lens):
    g = torch.Generator().manual_seed(sum(lens))
    T = sum(lens)
    qkv = torch.randn(T, (nq + 2 * nkv) * d, generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    q = qkv[:, :nq * d].unflatten(-1, (nq, d))
    k = qkv[:, nq * d:(nq + nkv) * d].unflatten(-1, (nkv, d))
    v = qkv[:, (nq + nkv) * d:].unflatten(-1, (nkv, d))
    scale = d ** -0.5
    qd = qkv.to(DEV)
    out = ops.prefill_attention(qd[:, :nq * d].unflatten(-1, (nq, d)), qd[:, nq * d:(nq + nkv) * d].unflatten(-1, (nkv, d)),
                                qd[:, (nq + nkv) * d:].unflatten(-1, (nkv, d)), cu.to(DEV), cu.to(DEV), max(lens), scale)
    assert_prefill_close(out, q, k, v, cu, scale)


def test_prefill_attention_f16():
    """the f16 instantiation of the head-dim-128 flash kernel (v_mfma_f32_32x32x16_f16; one RNE-rounded f16 P per score): against the
    fp32-P oracle computed from the same f16 inputs -- f16 carries 3 more mantissa bits than bf16, so the single-P result sits
    well inside 1e-3"""
    nq, nkv, d, lens = 8, 2, 128, [190, 77]
    g = torch.Generator().manual_seed(7)
    T = sum(lens)
    qkv = torch.randn(T, (nq + 2 * nkv) * d, generator=g).half()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    q = qkv[:, :nq * d].unflatten(-1, (nq, d))
    k = qkv[:, nq * d:(nq + nkv) * d].unflatten(-1, (nkv, d))
    v = qkv[:, (nq + nkv) * d:].unflatten(-1, (nkv, d))
    scale = d ** -0.5
    qd = qkv.to(DEV)
    out = ops.prefill_attention(qd[:, :nq * d].unflatten(-1, (nq, d)), qd[:, nq * d:(nq + nkv) * d].unflatten(-1, (nkv, d)),
                                qd[:, (nq + nkv) * d:].unflatten(-1, (nkv, d)), cu.to(DEV), cu.to(DEV), max(lens), scale)
    ref = orc.attention_varlen(q, k, v, cu, cu, scale, causal=True)
    assert out.dtype == torch.float16 and torch.isfinite(out.float()).all()
    err = ((out.float().cpu() - ref.float()).norm() / ref.float().norm()).item()
    assert err <= 1e-3, err


@pytest.mark.parametrize("bs", [128, 16])
def test_chunked_prefill_bottom_right_causal(bs):
    B, nq, nkv, d = 3, 14, 2, 128
    kv_lens, q_lens = [300, 130, 64], [44, 130, 1]
    md, kc, vc, q = _paged_case(B, nq, nkv, d, bs, kv_lens, q_lens, torch.bfloat16, seed=bs)
    scale = d ** -0.5
    ref = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, causal=True)
    ref16 = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, causal=True,
                                p_round=True)
    out = ops.paged_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), md["q_cu_seq_lens"].to(DEV), md["kv_seq_lens"].to(DEV),
                              md["block_tables"].to(DEV), max(q_lens), max(kv_lens), scale, is_causal=True)
    from _bars import assert_p16_attention_close
    flash = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, causal=True,
                                p_round="flash") if bs % 64 == 0 else None     # (other page sizes: register-staged kernel, hi + lo P)
    assert_p16_attention_close(out, ref, ref16, flash)      # one 16-bit P on the LDS-DMA kernel (pages of 64 / 128 tokens)


def test_mlu_golden_vectors_through_hip():
    """The reference's own golden vectors (tests/core/layers/mlu/qwen2_attention_test.cpp:254-328)
    through the HIP kernels: bf16 matmul + RoPE + KV write + attention + o_proj."""
    import test_oracle_golden as G
    w = {k: v.to(DEV) for k, v in G._weights().items()}
    cache = orc.build_cos_sin_cache(2048, G.D, 1000000.0, torch.bfloat16).to(DEV)
    q_size, kv_size = G.NQ * G.D, G.NKV * G.D

    def layer(hidden, positions, kc, vc, slots, mode, **kw):
        qkv = ops.matmul(hidden, w["qkv"], w["qkv_b"])
        q, k, v = qkv[:, :q_size], qkv[:, q_size:q_size + kv_size], qkv[:, q_size + kv_size:]
        ops.rotary_embedding(positions, q, k, cache, True, head_size=G.D)
        k3, v3 = k.unflatten(-1, (G.NKV, G.D)), v.unflatten(-1, (G.NKV, G.D))
        ops.reshape_paged_cache(slots, k3, v3, kc, vc)
        scale = math.sqrt(1.0 / G.D)
        q3 = q.unflatten(-1, (G.NQ, G.D))
        if mode == "prefill":
            attn = ops.prefill_attention(q3, k3, v3, kw["cu"], kw["cu"], kw["max_len"], scale)
        else:
            attn = ops.paged_attention(q3, kc, vc, None, kw["kv_lens"], kw["table"], 1, kw["max_kv"], scale)
        return ops.matmul(attn, w["o"])

    B, S = 2, 128
    kc, vc = [t.to(DEV) for t in G._caches()]
    hidden = orc.make_noise(G.PFX + "prefill.hidden_states", (B * S, G.H), 0.02).to(DEV)
    per = G._block_num(S) * G.BS
    slots = torch.tensor([b * per + i for b in range(B) for i in range(S)], dtype=torch.int32, device=DEV)
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=DEV)
    out = layer(hidden, torch.arange(S, device=DEV).repeat(B), kc, vc, slots, "prefill", cu=cu, max_len=S)
    exp = [0.6796875, 0.67578125, 0.6875, 0.65625, 0.6640625, 0.6796875, 0.68359375, 0.67578125, 0.6796875, 0.66796875]
    G._assert_close_bf16(out.flatten()[:10].cpu(), exp, ulps=2)

    B, S = 4, 256
    kc, vc = [t.to(DEV) for t in G._caches()]
    hidden = orc.make_noise(G.PFX + "decode.hidden_states", (B, G.H), 0.02).to(DEV)
    kv = S + 1
    nblk = G._block_num(kv)
    per = nblk * G.BS
    slots = torch.tensor([b * per + (kv - 1) for b in range(B)], dtype=torch.int32, device=DEV)
    table = torch.arange(B * nblk, dtype=torch.int32, device=DEV).view(B, nblk)
    out = layer(hidden, torch.full((B,), S, device=DEV), kc, vc, slots, "decode",
                kv_lens=torch.full((B,), kv, dtype=torch.int32, device=DEV), table=table, max_kv=kv)
    exp = torch.tensor([0.0005264282, 0.0008239746, 0.0005722046, 0.0006027222, 0.000831604, 0.0004405975,
                        0.001037598, 0.001083374, 0.000289917, 0.0007820129])
    got = out.flatten()[:10].float().cpu()
    # round-4 review, weak #2: held to <= 2 bf16 ulp per value like the prefill golden above (was: 3 % / 4e-5 absolute, which
    # pinned nothing on values of 3e-4 .. 1e-3). The golden comes from an MLU kernel that rounds P to bf16 before PV (the oracle
    # reproduces it to the last digit only in that mode, tests/test_oracle_golden.py::test_decode_golden); this kernel keeps P
    # to ~2^-17 (hi + lo parts), so it sits where the oracle with fp32 P sits: 2.4e-3 (relative L2) from the golden, inside
    # 2 ulp on every value -- and within 1e-3 / 2 ulp of THAT oracle on the whole output
    G._assert_close_bf16(got, exp.tolist(), ulps=2)
    e_gold = ((got - exp).norm() / exp.norm()).item()
    kc_o, vc_o = G._caches()
    ref = G._layer(hidden.cpu(), torch.full((B,), S), G._weights(), kc_o, vc_o, slots.cpu(), "decode", p_round=False,
                   cu_q=torch.arange(B + 1, dtype=torch.int32), kv_lens=torch.full((B,), kv, dtype=torch.int32),
                   block_table=table.cpu())
    e_ref = ((ref.flatten()[:10].float() - exp).norm() / exp.norm()).item()
    assert e_gold <= max(1.5 * e_ref, 3e-3), (e_gold, e_ref)
    assert rel_l2(out.cpu(), ref) <= 2e-3
    assert ((out.float().cpu() - ref.float()).abs() <= 2 * 2.0 ** -8 * ref.float().abs().amax(-1, keepdim=True)).all()

    # MixedSequenceLengthTest (:331-393): the reference's only vector for ragged q_cu_seq_lens -- 32 / 64 / 128 tokens in one
    # varlen prefill -- through the same HIP kernels, and equal to the oracle layer on EVERY output row (not only the ten values)
    hidden_c, positions_c, slots_c, cu_c = G._mixed_inputs()
    kc, vc = [t.to(DEV) for t in G._caches()]
    out = layer(hidden_c.to(DEV), positions_c.to(DEV), kc, vc, slots_c.to(DEV), "prefill", cu=cu_c.to(DEV), max_len=128)
    G._assert_close_bf16(out.flatten()[:10].cpu(), G.GOLD["qwen2_attention_mixed"]["first10"], ulps=2)
    kc_o, vc_o = G._caches()
    ref = G._layer(hidden_c, positions_c, G._weights(), kc_o, vc_o, slots_c, "prefill", cu=cu_c)
    assert rel_l2(out.cpu(), ref) <= 2e-3
    # the KV rows written to slots 0 .. 223 (a 16-bit GEMM in another summation order: equal to the last bit or so)
    assert_ulp_close(kc, kc_o, torch.bfloat16, ulps=2.0, min_exact=0.9)
    assert_ulp_close(vc, vc_o, torch.bfloat16, ulps=2.0, min_exact=0.9)


@pytest.mark.parametrize("M,N,K,dtype", [(64, 3584, 18944, torch.bfloat16), (64, 4608, 3584, torch.bfloat16),
                                         (256, 3584, 3584, torch.float16), (7, 1000, 8192, torch.bfloat16),
                                         (2048, 128, 2048, torch.bfloat16), (1100, 72, 1024, torch.float16)])
def test_matmul_16bit_split_k_is_deterministic_and_keeps_the_workspace_zero(M, N, K, dtype):
    """decode-shaped 16-bit GEMMs -- and tall ones with few columns (MoE router gates: few 128 x 128 tiles, round 6) -- split K
    through fp32 slabs in the workspace: == oracle (<= 1 ulp), identical bits run to run, and the workspace is zero again
    afterwards (an int8 split-K GEMM right after stays exact)"""
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, generator=g).to(dtype)
    ref = orc.matmul(a, w, bias)
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    out1 = ops.matmul(ad, wd, bd)
    out2 = ops.matmul(ad, wd, bd)
    assert torch.equal(out1, out2)
    assert_ulp_close(out1, ref, dtype, ulps=1.0, min_exact=0.97)
    ai = torch.randint(-127, 128, (M, 3584), dtype=torch.int8, device=DEV)
    wi = torch.randint(-127, 128, (256, 3584), dtype=torch.int8, device=DEV)
    ones_m, ones_n = torch.ones(M, device=DEV), torch.ones(256, device=DEV)
    acc = torch.empty(M, 256, dtype=torch.int32, device=DEV)
    ops.scaled_matmul(ai, wi, ones_m, ones_n, torch.bfloat16, acc_out=acc)          # exact accumulators (no split)
    y = ops.scaled_matmul(ai, wi, ones_m, ones_n, torch.bfloat16)                   # planner's path (may split K)
    assert torch.equal(y, acc.float().bfloat16())


# ------------------------------------------------------------------------------------------- MLA / MoE
@pytest.mark.parametrize("H,bs,kv_lens", [(16, 64, [8192 // 8, 70, 1]), (128, 64, [300, 65]), (5, 16, [100]),
                                          (16, 128, [2048, 129, 128, 64, 63]), (16, 32, [1000, 33]),
                                          (24, 64, [4096])])
def test_mla_decode(H, bs, kv_lens):
    """flash_mla dense decode == softmax(scale q.K) K[:, :512] over the paged latent cache (oracle: the generic
    paged attention with nkv=1, d=576, dv=512, v_cache aliasing k_cache -- prefill_sdpa's formulation,
    layers/dcu/deepseek_v2_attention.cpp:212-262)"""
    B = len(kv_lens)
    g = torch.Generator().manual_seed(H + bs)
    pages = [(L + bs - 1) // bs for L in kv_lens]
    nb = sum(pages) + 2
    perm = torch.randperm(nb, generator=g).tolist()
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, [1] * B, blocks, bs)
    kc = torch.randn(nb, bs, 1, 576, generator=g).bfloat16()
    q = torch.randn(B, H, 576, generator=g).bfloat16()
    scale = (192 ** -0.5)
    ref = orc.paged_attention(q, kc, kc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, dv=512)
    out = ops.mla_decode(q.to(DEV), kc.to(DEV), md["kv_seq_lens"].to(DEV), md["block_tables"].to(DEV), 512, scale,
                         max(kv_lens))
    assert_attn_close(out.view(B, -1), ref)


@pytest.mark.parametrize("H,bs,lens", [(16, 64, [(40, 40), (1, 1), (97, 97)]), (8, 64, [(17, 300), (64, 64), (5, 133)]),
                                       (128, 16, [(33, 33)]),
                                       # enough tokens for the kernel that shares a KV tile between four query tokens:
                                       # groups that straddle sequences, a 1-token sequence, a tail group, chunked prefill
                                       (128, 64, [(130, 130), (1, 70), (3, 3), (66, 321), (2, 2), (51, 200)]),
                                       (24, 128, [(257, 257), (255, 600)])])
def test_mla_prefill_and_latent_store(H, bs, lens):
    """a6: store_latent_cache (K-only cache write, bit-exact) then prefill / chunked prefill over the paged latent
    cache == softmax(scale q.K^T) K[:, :512] with the bottom-right causal mask (prefill_sdpa's formulation,
    layers/dcu/deepseek_v2_attention.cpp:212-262; oracle = generic paged attention, nkv=1, d=576, dv=512)"""
    q_lens, kv_lens = [a for a, _ in lens], [b for _, b in lens]
    B, T = len(lens), sum(q_lens)
    g = torch.Generator().manual_seed(H * 7 + bs)
    pages = [(L + bs - 1) // bs for L in kv_lens]
    nb = sum(pages) + 3
    perm = torch.randperm(nb, generator=g).tolist()
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, q_lens, blocks, bs)
    kc = torch.randn(nb, bs, 1, 576, generator=g).bfloat16()          # cached prefix (+ garbage elsewhere)
    latent = torch.randn(T, 576, generator=g).bfloat16()               # this step's [c_kv_normed || k_pe] rows
    kc_ref = kc.clone()
    orc.reshape_paged_cache(md["new_cache_slots"], latent.view(T, 1, 576), None, kc_ref, None)
    kc_dev = kc.to(DEV)
    ops.store_latent_cache(latent.to(DEV), md["new_cache_slots"].to(DEV), kc_dev)
    assert torch.equal(kc_dev.cpu().view(torch.int16), kc_ref.view(torch.int16))
    q = torch.randn(T, H, 576, generator=g).bfloat16()
    scale = 192 ** -0.5
    ref = orc.paged_attention(q, kc_ref, kc_ref, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                              causal=True, dv=512)
    out = ops.mla_prefill(q.to(DEV), kc_dev, md["q_cu_seq_lens"].to(DEV), md["kv_seq_lens"].to(DEV),
                          md["block_tables"].to(DEV), 512, scale, max(kv_lens), is_causal=True)
    # the tile-sharing kernel keeps ONE 16-bit P per score like prefill_sdpa (torch SDPA) does: bars relative to the p_round
    # oracle (tests/_bars.py); the per-token path (decode kernel, P = hi + lo) meets the same bars with room to spare
    from _bars import assert_p16_attention_close
    ref16 = orc.paged_attention(q, kc_ref, kc_ref, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                                causal=True, dv=512, p_round=True)
    # (the tile-sharing kernel needs pages that are a multiple of its 64-token tiles; other page sizes run per token, P = hi + lo)
    shared = bs % 64 == 0 and not (T * ((H + 15) // 16) < 4 * 128 and os.environ.get("XLLM_MI355_MLA_PREFILL", "") != "1") and \
        os.environ.get("XLLM_MI355_MLA_PREFILL", "") != "0" and os.environ.get("XLLM_MI355_MLA_PREFILL_P", "") != "2"
    flash = orc.paged_attention(q, kc_ref, kc_ref, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                                causal=True, dv=512, p_round="flash") if shared else None   # the tile-sharing kernel's cast point
    assert_p16_attention_close(out.view(T, -1), ref.view(T, -1), ref16.view(T, -1), None if flash is None else flash.view(T, -1))
    if T >= 256:                                                       # the unmasked form of the same entry
        ref_nc = orc.paged_attention(q, kc_ref, kc_ref, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                                     causal=False, dv=512)
        ref_nc16 = orc.paged_attention(q, kc_ref, kc_ref, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                                       causal=False, dv=512, p_round=True)
        out_nc = ops.mla_prefill(q.to(DEV), kc_dev, md["q_cu_seq_lens"].to(DEV), md["kv_seq_lens"].to(DEV),
                                 md["block_tables"].to(DEV), 512, scale, max(kv_lens), is_causal=False)
        assert_p16_attention_close(out_nc.view(T, -1), ref_nc.view(T, -1), ref_nc16.view(T, -1))
    # decode through the same cache agrees with the last query row of every sequence
    dec = ops.mla_decode(q[md["q_cu_seq_lens"][1:].long() - 1].contiguous().to(DEV), kc_dev, md["kv_seq_lens"].to(DEV),
                         md["block_tables"].to(DEV), 512, scale, max(kv_lens))
    last = out.cpu()[md["q_cu_seq_lens"][1:].long() - 1]
    if T * ((H + 15) // 16) < 4 * 128 and os.environ.get("XLLM_MI355_MLA_PREFILL", "") != "1":
        assert torch.equal(dec.cpu(), last)             # the per-token path IS the decode kernel: bit-equal
    else:                                               # tile-sharing kernel: one 16-bit P (decode: hi + lo), another sum order
        assert_attn_close(dec.view(B, -1), last.view(B, -1), rel=6e-3)


def test_moe_index_combine_group_gemm():
    T, topk, E, Hd, N = 301, 8, 128, 256, 384
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(T, E, generator=g)
    w, ids = logits.softmax(-1).topk(topk, -1)
    ids = ids.to(torch.int32)
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.to(DEV), E)
    r_src_dst, r_dst_src, r_sizes = orc.moe_compute_index(ids, E)
    assert torch.equal(sizes.cpu(), r_sizes)            # expert sizes: exact
    assert torch.equal(src_dst.cpu(), r_src_dst)        # our placement is stable => equals the oracle's order
    assert torch.equal(dst_src.cpu(), r_dst_src)
    # grouped GEMM on the sorted rows (bf16, reference DCU path) vs per-expert oracle matmul
    x = torch.randn(T, Hd, generator=g).bfloat16()
    xs = x[(dst_src.cpu().long() // topk)]              # expand + sort (index_select in fused_moe.cpp:250-262)
    we = (torch.randn(E, N, Hd, generator=g) / 16).bfloat16()
    got = ops.group_gemm(xs.to(DEV), we.to(DEV), sizes)
    ref = orc.group_gemm(xs, we, r_sizes)
    assert_ulp_close(got, ref, torch.bfloat16, ulps=2.0, min_exact=0.9)
    # combine (permutation-invariant result)
    g2 = torch.randn(T * topk, Hd, generator=g).bfloat16()
    outc = ops.moe_combine_result(g2.to(DEV), w.to(DEV), T, topk)
    refc = orc.moe_combine(g2, w.contiguous(), T, topk)
    assert_ulp_close(outc, refc, torch.bfloat16, ulps=1.0, min_exact=0.99)
    # fused un-sort + combine == index_copy_ + combine, bit for bit
    g2_sorted = torch.empty_like(g2).to(DEV)
    g2_sorted[src_dst.long()] = g2.to(DEV)
    assert torch.equal(ops.moe_combine_sorted(g2_sorted, src_dst, w.to(DEV), T, topk), outc)


@pytest.mark.parametrize("T,topk,E,K,N", [(1024, 8, 128, 2048, 1536), (777, 8, 128, 768, 2048), (300, 2, 8, 512, 264),
                                          (1024, 4, 5, 256, 512)])
def test_moe_group_gemm_tile_table_path(T, topk, E, K, N):
    """grouped GEMM on the 256x256 kernel behind the device-built tile table (cfg5 shapes, ragged / empty experts, N
    that is not a tile multiple) == per-expert oracle matmul; index build == oracle on the same ids (stable order)"""
    g = torch.Generator().manual_seed(T + E)
    logits = torch.randn(T, E, generator=g)
    if E == 5:
        logits[:, 3] = -1e9                              # an expert that receives no token at all
    ids = logits.topk(topk, -1).indices.to(torch.int32)
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.to(DEV), E)
    r_src_dst, r_dst_src, r_sizes = orc.moe_compute_index(ids, E)
    assert torch.equal(sizes.cpu(), r_sizes) and torch.equal(src_dst.cpu(), r_src_dst) and torch.equal(dst_src.cpu(), r_dst_src)
    x = torch.randn(T, K, generator=g).bfloat16()
    xs = x[(r_dst_src.long() // topk)]
    we = (torch.randn(E, N, K, generator=g) / math.sqrt(K)).bfloat16()
    got = ops.group_gemm(xs.to(DEV), we.to(DEV), sizes)
    ref = orc.group_gemm(xs, we, r_sizes)
    assert_ulp_close(got, ref, torch.bfloat16, ulps=2.0, min_exact=0.9)
    fused = ops.group_gemm_gather(x.to(DEV), dst_src, topk, we.to(DEV), sizes)   # expand fused into the A staging
    assert fused is None or torch.equal(fused, got)
    assert fused is not None or T * topk < max(1024, 64 * E) or os.environ.get("XLLM_MI355_GROUP_P8") == "0"


@pytest.mark.parametrize("T,topk,E,K,N", [(1024, 8, 128, 2048, 1536), (700, 4, 16, 768, 2048), (1200, 2, 5, 256, 520),
                                          (900, 2, 256, 256, 256), (300, 1, 1, 512, 264)])
def test_moe_group_gemm_w8a8(T, topk, E, K, N):
    """W8A8 grouped GEMM on the int8 8-phase kernel == the oracle's scaled_matmul per expert, bit for bit (exact int32 sums,
    same fp32 dequant expression); the gather form (each token quantised once, expanded inside the A staging) == the
    expanded form"""
    g = torch.Generator().manual_seed(T + E + N)
    logits = torch.randn(T, E, generator=g)
    if E == 5:
        logits[:, 2] = -1e9
    ids = logits.topk(topk, -1).indices.to(torch.int32)
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.to(DEV), E)
    r_sizes = sizes.cpu()
    x = torch.randn(T, K, generator=g).bfloat16()
    xq, xs = orc.scaled_quantize(x)
    wq = torch.randint(-127, 128, (E, N, K), generator=g, dtype=torch.int8)
    ws = torch.rand(E, N, generator=g) * 0.02 + 0.001
    tok = (dst_src.cpu().long() // topk)
    xq_s, xs_s = xq[tok], xs[tok]
    ref = torch.empty(T * topk, N, dtype=torch.bfloat16)
    off = 0
    for e in range(E):
        c = int(r_sizes[e])
        if c:
            ref[off:off + c] = orc.scaled_matmul(xq_s[off:off + c].contiguous(), wq[e], xs_s[off:off + c].contiguous(), ws[e], torch.bfloat16, None)
        off += c
    got = ops.group_gemm_w8a8(xq_s.to(DEV), xs_s.to(DEV), wq.to(DEV), ws.to(DEV), sizes)
    assert torch.equal(got.cpu(), ref)
    fused = ops.group_gemm_w8a8(xq.to(DEV), xs.to(DEV), wq.to(DEV), ws.to(DEV), sizes, row_index=dst_src, index_div=topk)
    assert torch.equal(fused, got)


def test_moe_group_gemm_w8a8_beyond_the_inline_plan():
    """more than 256 experts: the 8-phase kernels take their slots from the table group_plan_kernel builds (round 6: up to 256
    experts every workgroup plans for itself, no plan launch) -- the same bits as the oracle's scaled_matmul per expert, with
    empty experts, one-row experts and experts that straddle a 256-row tile"""
    E, K, N = 300, 256, 256
    g = torch.Generator().manual_seed(E)
    sizes = torch.randint(40, 130, (E,), generator=g, dtype=torch.int32)
    sizes[[0, 7, 150, 299]] = 0
    sizes[[3, 200]] = 1
    sizes[100] = 700
    rows = int(sizes.sum())
    xq = torch.randint(-127, 128, (rows, K), generator=g, dtype=torch.int8)
    xs = torch.rand(rows, generator=g) * 0.02 + 0.001
    wq = torch.randint(-127, 128, (E, N, K), generator=g, dtype=torch.int8)
    ws = torch.rand(E, N, generator=g) * 0.02 + 0.001
    got = ops.group_gemm_w8a8(xq.to(DEV), xs.to(DEV), wq.to(DEV), ws.to(DEV), sizes.to(DEV)).cpu()
    off = 0
    for e in range(E):
        c = int(sizes[e])
        if c and e in (1, 3, 99, 100, 101, 200, 298):      # sampled experts (the oracle's GEMM on every one of 300 is slow)
            ref = orc.scaled_matmul(xq[off:off + c].contiguous(), wq[e], xs[off:off + c].contiguous(), ws[e], torch.bfloat16, None)
            assert torch.equal(got[off:off + c], ref), e
        off += c
    # ... and against the inline plan on a sub-problem of the same data (the first 200 experts)
    r200 = int(sizes[:200].sum())
    sub = ops.group_gemm_w8a8(xq[:r200].to(DEV), xs[:r200].to(DEV), wq[:200].to(DEV), ws[:200].to(DEV), sizes[:200].to(DEV)).cpu()
    assert torch.equal(sub, got[:r200])


def test_fused_moe_layer_matches_dense_reference_and_unfused_operators():
    """FusedMoE.forward_experts (fused_moe.cpp:217-337): (a) fused expand / un-sort == the reference operator sequence,
    bit for bit; (b) == a dense per-token evaluation of the selected experts with the same rounding points
    (16-bit GEMM outputs, 16-bit SiLU * mul, fp32 weighted sum)"""
    from xllm_amd import layers
    T, H, I, E, topk = 1200, 512, 384, 16, 4
    gd = torch.Generator(device=DEV).manual_seed(3)
    moe = layers.FusedMoE(H, I, E, topk, torch.bfloat16, DEV, gd)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    logits = torch.randn(T, E, device=DEV, generator=gd).bfloat16()
    out = moe.forward_experts(x, logits)
    moe.fuse = False
    assert torch.equal(moe.forward_experts(x, logits), out)
    w, ids = ops.moe_fused_topk(logits, topk, True)
    xf = x.float()
    ref = torch.zeros(T, H, device=DEV)
    for k in range(topk):
        e = ids[:, k].long()
        h13 = torch.einsum("th,tnh->tn", xf, moe.w13[e].float()).bfloat16().float()
        act = (torch.nn.functional.silu(h13[:, :I]).bfloat16().float() * h13[:, I:]).bfloat16().float()
        ref += w[:, k, None] * torch.einsum("ti,thi->th", act, moe.w2[e].float()).bfloat16().float()
    got, ref = out.float(), ref.bfloat16().float()
    assert ((got - ref).norm() / ref.norm()).item() <= 4e-3


def test_fused_moe_layer_w8a8_tracks_the_16bit_layer():
    """FusedMoE(mode="int8"): W8A8 experts (per-token activation scales, per-channel expert weight scales) stay within the
    quantisation error of the 16-bit layer on the same weights and routing, and are reproducible bit for bit"""
    from xllm_amd import layers
    T, H, I, E, topk = 2048, 512, 384, 16, 4
    mk = lambda mode: layers.FusedMoE(H, I, E, topk, torch.bfloat16, DEV, torch.Generator(device=DEV).manual_seed(3), mode=mode)
    m16, m8 = mk("16bit"), mk("int8")
    gd = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    logits = torch.randn(T, E, device=DEV, generator=gd).bfloat16()
    ref, out = m16.forward_experts(x, logits), m8.forward_experts(x, logits)
    assert torch.equal(out, m8.forward_experts(x, logits))
    assert ((out.float() - ref.float()).norm() / ref.float().norm()).item() <= 4e-2


@pytest.mark.parametrize("mode,ep", [("16bit", 2), ("16bit", 4), ("int8", 4)])
def test_fused_moe_expert_parallel_ranks_add_up(mode, ep):
    """EP as in the reference's DCU layer (fused_moe.cpp:53-63, 236-315): a rank computes only its own experts, its output
    holds zeros for the others, and the EP all-reduce (here: the sum over the ranks computed one after another on one GPU)
    gives the all-experts layer. Per rank: fused path (ids rotated so the local experts sort first, combine skips the
    rows of the other ranks, no host sync) == the reference's own sequence (zeros + index_copy_ of the local rows +
    combine) bit for bit, also from a captured graph; the grouped gate and the shared experts ride along."""
    from xllm_amd import layers
    T, H, I, E, topk = 1024, 512, 384, 16, 4
    mk = lambda **kw: layers.FusedMoE(H, I, E, topk, torch.bfloat16, DEV, torch.Generator(device=DEV).manual_seed(11),
                                      mode=mode, num_expert_group=4, topk_group=2, scoring_func="sigmoid",
                                      correction_bias=torch.linspace(-0.1, 0.1, E, device=DEV), route_scale=2.5, **kw)
    gd = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    logits = torch.randn(T, E, device=DEV, generator=gd).bfloat16()
    full = mk().forward_experts(x, logits).float()
    total = torch.zeros_like(full)
    for r in range(ep):
        rank = mk(ep_rank=r, ep_size=ep)
        assert rank.w13.size(0) == E // ep
        part = rank.forward_experts(x, logits)
        if mode == "16bit":
            rank.fuse = False
            assert torch.equal(rank.forward_experts(x, logits), part)
            rank.fuse = True
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            captured = rank.forward_experts(x, logits)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(captured, part)
        total += part.float()
    # every (token, expert) product is computed by exactly one rank with the same arithmetic; only the 16-bit rounding of
    # the per-rank partial sums differs from the single-rank layer
    assert ((total - full).norm() / full.norm()).item() <= 6e-3
    shared = lambda t: (t.float() * 0.5).to(t.dtype)
    with_shared = mk(ep_rank=0, ep_size=ep, shared_experts=shared).forward_experts(x, logits)
    assert torch.equal(with_shared, mk(ep_rank=0, ep_size=ep).forward_experts(x, logits) + shared(x))


@pytest.mark.parametrize("mode", ["16bit", "int8"])
def test_fused_moe_alltoall_path_on_one_rank_equals_the_layer(mode):
    """N4 all-to-all EP (parallel.ep_dispatch / ep_combine + forward_experts_alltoall): with one rank the exchange is the
    identity, so the dispatched rows, the expert-local index build (topk = 1 over the arrival order), the grouped GEMMs
    and the un-sort must reproduce forward_experts bit for bit (the 2-rank exchange itself runs in test_parallel_gloo.py)"""
    from xllm_amd import layers
    T, H, I, E, topk = 900, 512, 384, 16, 4
    moe = layers.FusedMoE(H, I, E, topk, torch.bfloat16, DEV, torch.Generator(device=DEV).manual_seed(21), mode=mode,
                          num_expert_group=4, topk_group=3)
    gd = torch.Generator(device=DEV).manual_seed(22)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    logits = torch.randn(T, E, device=DEV, generator=gd).bfloat16()
    assert torch.equal(moe.forward_experts_alltoall(x, logits), moe.forward_experts(x, logits))


@pytest.mark.parametrize("mode", ["16bit", "int8"])
def test_fused_moe_layer_replays_from_a_hip_graph(mode):
    """the whole expert path (top-k, index build, tile-table plan, gathered grouped GEMMs, combine) has no host sync: one
    captured graph, replayed on NEW inputs written into the static buffers, equals the eager layer bit for bit (the
    reference's DCU group_gemm reads token_count on the host, kernels/dcu/group_gemm.cpp:45, and cannot be captured)"""
    from xllm_amd import layers
    T, H, I, E, topk = 1536, 512, 384, 16, 4
    moe = layers.FusedMoE(H, I, E, topk, torch.bfloat16, DEV, torch.Generator(device=DEV).manual_seed(5), mode=mode)
    gd = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    logits = torch.randn(T, E, device=DEV, generator=gd).bfloat16()
    moe.forward_experts(x, logits)                      # warm-up: workspaces are registered outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = moe.forward_experts(x, logits)
    x.copy_(torch.randn(T, H, device=DEV, generator=gd).bfloat16())
    logits.copy_(torch.randn(T, E, device=DEV, generator=gd).bfloat16())   # new routing, new expert sizes
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, moe.forward_experts(x, logits))


# ------------------------------------------------------------------------------------------- N1 fusions
@pytest.mark.parametrize("d,mode,T", [(18944, "silu", 7), (4864, "silu", 7), (32000, "silu", 7), (1024, "gelu", 7),
                                       (768, "silu", 3001), (512, "silu", 600), (1024, "silu", 1000), (96, "silu", 513)])
def test_act_and_mul_int8_fusion_equals_two_ops(d, mode, T):
    """(the short-row / many-row cases run the one-wave-per-row kernel used for MoE expert widths)"""
    g = torch.Generator().manual_seed(d)
    x = (torch.randn(T, 2 * d, generator=g) * 2).bfloat16().to(DEV)
    act = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    ops.act_and_mul(act, x, mode)
    q_ref, s_ref = ops.scaled_quantize(act)
    q, s = ops.act_and_mul_dynamic_int8_quant(x, mode)
    assert torch.equal(q, q_ref) and torch.equal(s, s_ref)


@pytest.mark.parametrize("nq,nkv,bs", [(28, 4, 128), (14, 2, 128), (7, 1, 128), (14, 2, 16)])
def test_decode_fusions_equal_unfused_ops(nq, nkv, bs):
    """rotary_embedding_and_cache == rotary_embedding + reshape_paged_cache, and paged_decode_attention_int8 ==
    paged_attention + scaled_quantize, bit for bit"""
    B, d = 256, 128
    kv_lens = [257 + 3 * i for i in range(B)]
    md, kc, vc, _ = _paged_case(B, nq, nkv, d, bs, kv_lens, [1] * B, torch.bfloat16, seed=nq)
    g = torch.Generator().manual_seed(nq + 100)
    qkv = torch.randn(B, (nq + 2 * nkv) * d, generator=g).bfloat16()
    cache = orc.build_cos_sin_cache(2048, d, 1e6, torch.bfloat16).to(DEV)
    pos = torch.tensor([L - 1 for L in kv_lens], device=DEV)  # int64
    slots = md["new_cache_slots"].to(DEV)
    kv_d, bt = md["kv_seq_lens"].to(DEV), md["block_tables"].to(DEV)

    def split(t):
        return t[:, :nq * d], t[:, nq * d:(nq + nkv) * d], t[:, (nq + nkv) * d:]

    a = qkv.to(DEV); kc_a, vc_a = kc.to(DEV), vc.to(DEV)
    qa, ka, va = split(a)
    ops.rotary_embedding(pos, qa, ka, cache, True, head_size=d)
    ops.reshape_paged_cache(slots, ka.unflatten(-1, (nkv, d)), va.unflatten(-1, (nkv, d)), kc_a, vc_a)
    out_a = ops.paged_attention(qa.unflatten(-1, (nq, d)), kc_a, vc_a, None, kv_d, bt, 1, max(kv_lens), d ** -0.5)
    qa_q, qa_s = ops.scaled_quantize(out_a)

    b = qkv.to(DEV); kc_b, vc_b = kc.to(DEV), vc.to(DEV)
    qb, kb, vb = split(b)
    ops.rotary_embedding_and_cache(pos, qb, kb, vb, cache, slots, kc_b, vc_b, d, True)
    assert torch.equal(a, b) and torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    fused = ops.paged_decode_attention_int8(qb.unflatten(-1, (nq, d)), kc_b, vc_b, kv_d, bt, max(kv_lens), d ** -0.5,
                                            want_16bit=True)
    assert fused is not None
    assert torch.equal(fused[2], out_a) and torch.equal(fused[0], qa_q) and torch.equal(fused[1], qa_s)


@pytest.mark.parametrize("B,S,nq,nkv,must_fuse", [(2, 4096, 28, 4, True), (32, 4096, 28, 4, True), (8, 3000, 28, 4, True),
                                                  (64, 300, 28, 4, False), (128, 257, 28, 4, False), (5, 2000, 14, 2, False),
                                                  (3, 2500, 7, 1, True)])
def test_decode_int8_fusion_on_split_plans_equals_the_two_operators(B, S, nq, nkv, must_fuse):
    """small batches: when the token range is split over the grid, the partials go through ONE merge + quantise launch -- the
    bits of paged_attention followed by scaled_quantize (plans that only split inside the workgroup are declined: nothing to
    gain there, the caller runs the two operators)"""
    kv_lens = [max(1, S - 13 * i) for i in range(B)]
    md, kc, vc, q = _paged_case(B, nq, nkv, 128, 128, kv_lens, [1] * B, torch.bfloat16, seed=B)
    qd, kcd, vcd = q.to(DEV), kc.to(DEV), vc.to(DEV)
    kv_d, bt = md["kv_seq_lens"].to(DEV), md["block_tables"].to(DEV)
    ref = ops.paged_attention(qd, kcd, vcd, None, kv_d, bt, 1, max(kv_lens), 128 ** -0.5)
    rq, rs = ops.scaled_quantize(ref)
    r = ops.paged_decode_attention_int8(qd, kcd, vcd, kv_d, bt, max(kv_lens), 128 ** -0.5, want_16bit=True)
    if os.environ.get("XLLM_MI355_ATTN_FINISH", "1") != "1":
        assert r is None or torch.equal(r[0], rq)        # switched off: decline, the caller runs the two operators
        return
    assert r is not None or not must_fuse
    if r is None:
        return
    assert torch.equal(r[2], ref) and torch.equal(r[0], rq) and torch.equal(r[1], rs)
    r2 = ops.paged_decode_attention_int8(qd, kcd, vcd, kv_d, bt, max(kv_lens), 128 ** -0.5)     # without the 16-bit copy
    assert torch.equal(r2[0], rq) and torch.equal(r2[1], rs) and r2[2] is None


@pytest.mark.parametrize("M,nq,nkv", [(1, 28, 4), (32, 28, 4), (100, 28, 4), (128, 14, 2), (64, 7, 1)])
def test_qkv_gemm_rope_cache_fusion_equals_the_three_operators(M, nq, nkv):
    """packed W8A8 qkv projection -> dequant -> RoPE -> KV write in two launches == scaled_matmul + rotary_embedding +
    reshape_paged_cache, bit for bit (qkv rows and both caches); skipped slots (-1) are not written"""
    d, K, bs = 128, 3584, 128
    N = (nq + 2 * nkv) * d
    g = torch.Generator().manual_seed(M + nq)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV)
    wp = ops.pack_weight_i8(w)
    cache = orc.build_cos_sin_cache(4096, d, 1e6, torch.bfloat16).to(DEV)
    pos = torch.randint(0, 4096, (M,), generator=g).to(DEV)
    nb = (M + bs - 1) // bs + 2
    slots = torch.randperm(nb * bs, generator=g)[:M].to(torch.int32)
    if M > 2:
        slots[1] = -1
    slots = slots.to(DEV)
    kc_a, vc_a = [torch.randn(nb, bs, nkv, d, generator=g).bfloat16().to(DEV) for _ in range(2)]
    kc_b, vc_b = kc_a.clone(), vc_a.clone()
    ref = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias, b_packed=wp)
    q, k, v = ref[:, :nq * d], ref[:, nq * d:(nq + nkv) * d], ref[:, (nq + nkv) * d:]
    ops.rotary_embedding_and_cache(pos, q, k, v, cache, slots, kc_a, vc_a, d, True)
    got = ops.scaled_matmul_rope_cache(a, wp, a_s, w_s, bias, pos, cache, slots, kc_b, vc_b, nq, nkv, d)
    if os.environ.get("XLLM_MI355_QKV_ROPE", "1") == "0" or os.environ.get("XLLM_MI355_PACKED", "auto") == "0":
        assert got is None        # the fusion is switched off / no GEMM runs on packed weights: the three operators serve
        return
    assert got is not None
    assert torch.equal(got, ref) and torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)


def test_piecewise_graph_replay_equals_eager():
    """N2: a TP step replayed as [graph | collective | graph | ...] (xllm_amd.parallel.PiecewiseGraph) produces the
    bits of the eager step. The "collective" here is a local stand-in with a visible effect (the real RCCL / gloo
    exchange runs outside every capture exactly like this callable does)."""
    import bench
    from xllm_amd import layers, parallel
    from xllm_amd.attention import KVCache

    calls = {"n": 0}

    class FakeTP(parallel.ProcessGroup):
        def allreduce(self, x):
            def fn():
                calls["n"] += 1
                x.mul_(2.0)          # what a 2-rank SUM of identical shards would do
            parallel._run_collective(fn)

        def allgather(self, x):
            out = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            xc = x.contiguous()

            def fn():
                calls["n"] += 1
                out[0].copy_(xc)
                out[1].copy_(xc)
            parallel._run_collective(fn)
            return out

    args = layers.ModelArgs(1024, 3, 16, 4, 64, 2048, 4096, 1e-6, 1e6, 4096)  # every sharded K stays a multiple of 128
    B, ctx, bs = 32, 300, 128
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=3, tp=FakeTP(None, 0, 2), n_layers=3)
    md, n_blocks = bench.build_metadata(B, ctx, bs, torch.device(DEV), seed=1)
    g = torch.Generator(device=DEV).manual_seed(5)
    caches = [KVCache(torch.randn(n_blocks, bs, model.layers[0].nkv, args.head_dim, device=DEV, generator=g).bfloat16(),
                      torch.randn(n_blocks, bs, model.layers[0].nkv, args.head_dim, device=DEV, generator=g).bfloat16())
              for _ in model.layers]
    tokens = torch.randint(0, args.vocab_size, (B,), device=DEV, generator=g)
    positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=DEV)

    def step():
        return model.logits(model.forward(tokens, positions, md, caches))

    ref = step().clone()
    n_eager = calls["n"]
    assert n_eager == 2 * 3 + 1                      # two all-reduces per layer + the logits gather
    step()
    pw = parallel.PiecewiseGraph()
    static_out = pw.capture(step)
    assert sum(isinstance(it, torch.cuda.CUDAGraph) for it in pw.items) == n_eager + 1
    static_out.zero_()
    pw.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref)
    tokens.copy_(torch.randint(0, args.vocab_size, (B,), device=DEV, generator=g))   # new inputs, same buffers
    ref2 = step().clone()
    pw.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref2) and not torch.equal(ref2, ref)


@pytest.mark.parametrize("M,N,K,bias", [(256, 3584, 3584, False), (256, 3584, 18944, False), (37, 512, 1024, True),
                                        (300, 1024, 256, True)])
def test_gemm_add_norm_fusion_equals_separate_ops(M, N, K, bias):
    """N1 across the GEMM boundary: scaled_matmul_add_rms_norm == scaled_matmul -> fused_add_rms_norm (-> int8 quant),
    bit for bit, and it leaves the split-K workspace zeroed (the next plain GEMM is still exact)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV) if bias else None
    res0 = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    nw = (torch.rand(N, generator=g) + 0.5).bfloat16().to(DEV)
    # separate operators
    y = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b)
    res_ref = res0.clone()
    q_ref, s_ref = ops.rms_norm_dynamic_int8_quant(y.clone(), nw, 1e-6, residual=res_ref)
    res_ref2 = res0.clone()
    y2 = y.clone()
    ops.fused_add_rms_norm(y2, res_ref2, nw, 1e-6)      # y2 <- norm, res_ref2 <- y + residual
    # fused
    res_a = res0.clone()
    q, qs = ops.scaled_matmul_add_rms_norm(a, w, a_s, w_s, res_a, nw, 1e-6, b, quantize=True)
    assert torch.equal(q, q_ref) and torch.equal(qs, s_ref) and torch.equal(res_a, res_ref)
    res_b = res0.clone()
    n16 = ops.scaled_matmul_add_rms_norm(a, w, a_s, w_s, res_b, nw, 1e-6, b, quantize=False)
    assert torch.equal(n16, y2) and torch.equal(res_b, res_ref2)
    # workspace invariant: a following split-K GEMM is exact
    acc = torch.empty(M, N, dtype=torch.int32, device=DEV)
    ref = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b, acc_out=acc)
    assert torch.equal(ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b), ref)


def test_model_step_fused_equals_reference_operator_order():
    """every N1 fusion is bit-identical to the operators it replaces, so the whole decode step is too"""
    import bench
    from xllm_amd import layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(1024, 3, 16, 4, 128, 2048, 4096, 1e-6, 1e6, 4096)
    B, ctx, bs = 256, 300, 128
    outs = []
    for fuse in (True, False):
        model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=11, fuse=fuse, n_layers=3)
        md, n_blocks = bench.build_metadata(B, ctx, bs, torch.device(DEV), seed=2)
        g = torch.Generator(device=DEV).manual_seed(7)
        caches = [KVCache(torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16(),
                          torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16()) for _ in model.layers]
        tokens = torch.randint(0, args.vocab_size, (B,), device=DEV, generator=g)
        positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=DEV)
        outs.append(model.logits(model.forward(tokens, positions, md, caches)).clone())
    assert torch.equal(outs[0], outs[1])


def test_model_prefill_fused_equals_reference_operator_order():
    """the prefill chunk too: RoPE + KV write in one pass (16-byte accesses since round 4), norm + quant fused, gate_up with SiLU.mul
    in its epilogue -- hidden states and caches bit-identical to the reference operator order"""
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(1024, 3, 16, 4, 128, 2048, 4096, 1e-6, 1e6, 4096)
    bs, lens = 128, [1024, 768, 512]            # T = 2304
    pages = [L // bs for L in lens]
    table, used = [], 0
    for n in pages:
        table.append(list(range(used, used + n))); used += n
    bi = attention.build_batch_input([0] * len(lens), lens, table, bs)
    md = attention.build_attention_metadata(bi, is_prefill=True, is_chunked_prefill=False, device=torch.device(DEV))
    T = sum(lens)
    outs = []
    for fuse in (True, False):
        model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=11, fuse=fuse, n_layers=3)
        g = torch.Generator(device=DEV).manual_seed(7)
        caches = [KVCache(torch.zeros(used, bs, 4, 128, device=DEV, dtype=torch.bfloat16),
                          torch.zeros(used, bs, 4, 128, device=DEV, dtype=torch.bfloat16)) for _ in model.layers]
        tokens = torch.randint(0, args.vocab_size, (T,), device=DEV, generator=g)
        hidden = model.forward(tokens, bi.positions.to(DEV).long(), md, caches)
        outs.append((hidden.clone(), [c.k_cache.clone() for c in caches], [c.v_cache.clone() for c in caches]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
        assert torch.equal(a, b)


def _padded_decode_metadata(live_lens, B_pad, blocks, bs):
    """the persistent decode inputs of a bucket-padded graph (runtime/dcu_graph_executor_impl.cpp:166-332, 426-550): live rows as
    built for the batch; padding rows get token 0, position 0, slot -1, kv length 0 (extend_kv_cu_seq_lens_for_padding :59-70) and a
    zero-filled block-table row (:346, :529-531)"""
    from xllm_amd.attention import AttentionMetadata
    n_live = len(live_lens)
    md_o = orc.build_batch_metadata(list(live_lens), [1] * n_live, blocks, bs)
    max_blocks = md_o["block_tables"].size(1)
    kv = torch.zeros(B_pad, dtype=torch.int32)
    kv[:n_live] = md_o["kv_seq_lens"]
    slots = torch.full((B_pad,), -1, dtype=torch.int32)
    slots[:n_live] = md_o["new_cache_slots"]
    table = torch.zeros(B_pad, max_blocks, dtype=torch.int32)
    table[:n_live] = md_o["block_tables"]
    kv_cu = torch.zeros(B_pad + 1, dtype=torch.int32)
    kv_cu[1:] = torch.cumsum(kv, 0)
    md = AttentionMetadata(q_cu_seq_lens=torch.arange(B_pad + 1, dtype=torch.int32, device=DEV), kv_cu_seq_lens=kv_cu.to(DEV),
                           kv_seq_lens=kv.to(DEV), slot_mapping=slots.to(DEV), block_table=table.to(DEV), max_query_len=1,
                           max_seq_len=int(max(live_lens)), is_prefill=False, is_chunked_prefill=False)
    return md


@pytest.mark.parametrize("mode", ["int8", "bf16"])
def test_bucket_padded_graph_replay_live_rows_equal_the_unpadded_step(mode):
    """N2 (round-3 review, missing #5): the reference captures the decode graph at a padded batch and replays it with fewer live
    rows. Capture the B = 256 step, replay it with 200 live sequences (padding rows: slot -1, block-table row 0, token 0,
    kv length 0): live rows bit-equal to the UNPADDED B = 200 step, padded rows finite, no cache write outside the live slots,
    every zero-at-rest scratch zero afterwards. (Context 255 keeps both attention launch plans on one split, so bit-equality is
    the right bar; the split plans at long context are covered by the padded-row attention test below.)"""
    from xllm_amd import layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(1024, 2, 16, 4, 128, 2048, 4096, 1e-6, 1e6, 4096)
    B_pad, n_live, bs = 256, 200, 128
    g = torch.Generator().manual_seed(31)
    lens = [int(x) for x in torch.randint(1, 255, (n_live,), generator=g)]
    lens[0], lens[1], lens[-1] = 254, 1, 128
    pages = [(L + bs - 1) // bs for L in lens]
    nb = sum(pages) + 4
    perm = torch.randperm(nb, generator=g).tolist()
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=5, n_layers=2)

    def caches():
        gd = torch.Generator(device=DEV).manual_seed(77)
        return [KVCache(torch.empty(nb, bs, 4, 128, dtype=torch.bfloat16, device=DEV).normal_(generator=gd),
                        torch.empty(nb, bs, 4, 128, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)) for _ in model.layers]
    tok_live = torch.randint(0, args.vocab_size, (n_live,), generator=g)
    # (A) the unpadded step, eager
    md_a = _padded_decode_metadata(lens, n_live, blocks, bs)
    ca = caches()
    pos_a = torch.tensor([L - 1 for L in lens], dtype=torch.int64, device=DEV)
    ref_h = model.forward(tok_live.to(DEV), pos_a, md_a, ca).clone()
    ref = model.logits(ref_h).clone()
    # (B) the padded graph: captured on a FULL batch of other sequences, then its persistent buffers are refreshed in place
    md_b = _padded_decode_metadata(lens, B_pad, blocks, bs)
    cb = caches()
    before = [(c.k_cache.clone(), c.v_cache.clone()) for c in cb]
    tokens = torch.zeros(B_pad, dtype=torch.int64, device=DEV)
    positions = torch.zeros(B_pad, dtype=torch.int64, device=DEV)
    live_md = {k: getattr(md_b, k).clone() for k in ("kv_cu_seq_lens", "kv_seq_lens", "slot_mapping", "block_table")}
    # capture-time contents: every row live (the shape the graph was captured for), pointing at page 0 / slot of its own
    md_b.kv_seq_lens.fill_(1); md_b.slot_mapping.copy_(torch.arange(B_pad, dtype=torch.int32, device=DEV) % bs)
    md_b.block_table.zero_(); md_b.kv_cu_seq_lens.copy_(torch.arange(B_pad + 1, dtype=torch.int32, device=DEV))
    model.logits(model.forward(tokens, positions, md_b, cb))            # warm-up (scratch buffers exist before the capture)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        hid = model.forward(tokens, positions, md_b, cb)
        out = model.logits(hid)
    for c, (k0, v0) in zip(cb, before):                                  # undo the warm-up / capture-time cache writes
        c.k_cache.copy_(k0); c.v_cache.copy_(v0)
    for k, v in live_md.items():
        getattr(md_b, k).copy_(v)
    tokens.zero_(); positions.zero_()
    tokens[:n_live] = tok_live.to(DEV)
    positions[:n_live] = pos_a
    gr.replay()
    torch.cuda.synchronize()
    exact = mode == "int8"   # int8: exact int32 sums, row-wise consumers -> every row is independent of the batch it sits in.
    # 16-bit GEMMs (bf16 mode; the lm_head in both) sum in fp32 in an order the planner picks from M: one-ulp flips allowed there
    if exact:
        assert torch.equal(hid[:n_live], ref_h)                          # live rows: the unpadded step, bit for bit
    else:
        assert rel_l2(hid[:n_live], ref_h) <= 2e-3
    assert rel_l2(out[:n_live], ref) <= (1e-3 if exact else 3e-3)
    assert torch.isfinite(out.float()).all() and torch.isfinite(hid.float()).all()      # padded rows: finite
    # no cache write outside the live slots; the live slots hold what the unpadded step wrote
    for c_a, c_b, (k0, v0) in zip(ca, cb, before):
        if exact:
            assert torch.equal(c_b.k_cache, c_a.k_cache) and torch.equal(c_b.v_cache, c_a.v_cache)
        else:
            assert rel_l2(c_b.k_cache, c_a.k_cache) <= 1e-3 and rel_l2(c_b.v_cache, c_a.v_cache) <= 1e-3
        touched = torch.zeros(nb * bs, dtype=torch.bool, device=DEV)
        touched[md_a.slot_mapping.long()] = True
        assert torch.equal(c_b.k_cache.view(nb * bs, -1)[~touched], k0.view(nb * bs, -1)[~touched])
    # zero-at-rest scratch (row-amax of the gate_up fusion, split-K workspace) is zero again
    assert all(float(v.abs().max()) == 0.0 for v in ops._row_amax.values())
    assert all(int(v.view(torch.int32).abs().max()) == 0 for v in ops._gemm_ws.values())
    first = hid.clone()
    gr.replay()                                                          # and a second replay gives the same bits
    torch.cuda.synchronize()
    assert torch.equal(hid, first)


@pytest.mark.parametrize("B_pad,n_live,ctx", [(256, 200, 4096), (64, 40, 2048), (8, 3, 4096)])
def test_decode_attention_padded_rows_kv_len_zero(B_pad, n_live, ctx):
    """the attention operators on a padded batch at BASELINE's contexts (split-KV plans included: B = 64 / 8 split the token
    range over the grid, so the merge and the finishing kernels see (m, l) = (-big, 0) partials): rows with kv_len = 0 give
    ZEROS (l = 0 is pinned to 0, never 0 / 0; attention_decode.hip epilogues), live rows are bit-equal to the same launch
    without padding semantics (same batch size, padding rows given a real sequence), through paged_attention and through the
    fused int8 form."""
    nq, nkv, d, bs = 28, 4, 128, 128
    g = torch.Generator().manual_seed(B_pad + ctx)
    gd = torch.Generator(device=DEV).manual_seed(B_pad + ctx)
    pages = ctx // bs
    nb = B_pad * pages + 3
    table = torch.randperm(nb, generator=g)[:B_pad * pages].to(torch.int32).view(B_pad, pages).to(DEV)
    kc = torch.empty(nb, bs, nkv, d, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    vc = torch.empty(nb, bs, nkv, d, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    q = torch.empty(B_pad, nq, d, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    lens_full = torch.full((B_pad,), ctx, dtype=torch.int32, device=DEV)
    lens_full[0], lens_full[1] = 1, ctx - 7
    scale = 1.0 / math.sqrt(d)
    full = ops.paged_attention(q, kc, vc, None, lens_full, table, 1, ctx, scale)
    lens_pad = lens_full.clone()
    lens_pad[n_live:] = 0
    table_pad = table.clone()
    table_pad[n_live:] = 0
    out = ops.paged_attention(q, kc, vc, None, lens_pad, table_pad, 1, ctx, scale)
    assert torch.equal(out[:n_live], full[:n_live])
    assert int((out[n_live:] != 0).sum()) == 0
    fused = ops.paged_decode_attention_int8(q, kc, vc, lens_pad, table_pad, ctx, scale)
    if fused is not None:
        oq, osc = fused[0], fused[1]
        q_ref, s_ref = ops.scaled_quantize(out.view(B_pad, nq * d))
        assert torch.equal(oq.view(B_pad, -1), q_ref) and torch.equal(osc, s_ref)
        assert int(oq[n_live:].abs().max()) == 0 and float(osc[n_live:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["bf16", "int8"])
def test_model_prefill_chunked_prefill_and_decode_agree(mode):
    """end-to-end consistency of the three attention phases through the whole model (every row of a1-a15 in one property):
    the logits of token n+1 are the same whether the sequence was (a) prefilled in one shot, (b) prefilled up to a split
    point and continued by a CHUNKED prefill over the cached prefix, or (c) prefilled up to n and continued by a DECODE
    step -- the product's own host builder makes every batch; only the summation order differs, so the bar is 16-bit."""
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(512, 2, 8, 2, 64, 1024, 1000, 1e-6, 1e4, 4096)
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=9)
    bs, lens = 16, [37, 64, 5, 21]
    B = len(lens)
    g = torch.Generator().manual_seed(31)
    toks = [torch.randint(0, args.vocab_size, (L,), generator=g) for L in lens]
    need = [(L + bs - 1) // bs + 1 for L in lens]
    perm = torch.randperm(sum(need) + 2, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    nb = sum(need) + 2

    def caches():
        return [KVCache(torch.zeros(nb, bs, 2, 64, dtype=torch.bfloat16, device=DEV),
                        torch.zeros(nb, bs, 2, 64, dtype=torch.bfloat16, device=DEV)) for _ in model.layers]

    def run(cached, upto, kv, prefill, chunked):
        """feeds tokens [cached[b], upto[b]) of every sequence; returns the logits of each sequence's last fed token"""
        bi = attention.build_batch_input(cached, upto, blocks, bs)
        md = attention.build_attention_metadata(bi, prefill, chunked, DEV)
        ids = torch.cat([toks[b][cached[b]:upto[b]] for b in range(B)]).to(DEV)
        hidden = model.forward(ids, bi.positions.long().to(DEV), md, kv)
        last = (bi.q_cu_seq_lens[1:].long() - 1).to(DEV)
        return model.logits(hidden[last]).float()

    zero = [0] * B
    one_shot = run(zero, lens, caches(), True, False)                                  # (a)
    kv_b = caches()
    split = [L // 2 for L in lens]
    run(zero, split, kv_b, True, False)
    chunked = run(split, lens, kv_b, False, True)                                      # (b)
    kv_c = caches()
    run(zero, [L - 1 for L in lens], kv_c, True, False)
    decode = run([L - 1 for L in lens], lens, kv_c, False, False)                      # (c)
    for other in (chunked, decode):
        assert ((other - one_shot).norm() / one_shot.norm()).item() <= 2e-2
        assert torch.equal(other.argmax(-1), one_shot.argmax(-1)) or \
            ((other - one_shot).abs().max() / one_shot.abs().max()).item() <= 2e-2
    # the KV caches written by the three schedules hold the same rows. Layer 0 sees the same inputs in every schedule: with
    # int8 linears (exact integer sums whatever the kernel) its rows are bit-identical; with 16-bit linears a schedule that
    # feeds <= 64 tokens runs the weight-stream GEMM and one that feeds more the tiled kernels (another fp32 summation
    # order), so the rows agree to one 16-bit ulp
    ka, kb = kv_b[0].k_cache.float(), kv_c[0].k_cache.float()
    if mode == "int8":
        assert torch.equal(kv_b[0].k_cache, kv_c[0].k_cache)
    else:
        assert ((ka - kb).abs() <= 2.0 ** -7 * torch.maximum(ka.abs(), kb.abs()) + 1e-6).all()
        assert (ka == kb).float().mean().item() >= 0.9


# ------------------------------------------------------------------------------------------- N3 sampler
def test_philox_uniform_matches_oracle_and_hiprand(tmp_path):
    for seed, off in [(0, 0), (1234567, 5), (2 ** 63 + 11, 4 * 10 ** 9 + 3)]:
        assert torch.equal(ops.philox_uniform(3000, seed, off).cpu(), orc.philox_uniform(3000, seed, off))
    # pin against hiprand itself (header-only device API of the ROCm install) when a compiler is on the box
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hiprand_probe.hip")
    exe = tmp_path / "probe"
    try:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", str(exe), src], check=True, capture_output=True,
                       timeout=300)
    except (OSError, subprocess.SubprocessError):
        pytest.skip("no hipcc / hiprand headers on this box")
    vals = [float.fromhex(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True,
                                                     timeout=120).stdout.split()]
    got = torch.cat([ops.philox_uniform(512, 1234567, 5), ops.philox_uniform(512, 99, 4000000003)]).cpu()
    assert torch.equal(got, torch.tensor(vals, dtype=torch.float32))


@pytest.mark.parametrize("B,V", [(64, 152064), (33, 1003), (5, 8192)])
def test_random_sample(B, V):
    g = torch.Generator().manual_seed(B + V)
    probs = torch.softmax(torch.randn(B, V, generator=g) * 3, -1)
    probs[probs < 1e-6] = 0.0                      # top-p style zeros
    probs[1] = 0.0                                  # nothing valid -> 0
    u = torch.rand(B, generator=g)
    u[2] = 1.0                                      # beyond the total -> last valid index
    ref = orc.random_sample(probs, u)
    got = ops.random_sample(probs.to(DEV), uniform=u.to(DEV)).cpu()
    assert got[1] == 0 and got[2] == ref[2]
    same = (got == ref)
    assert same.float().mean() >= 0.95
    cdf = torch.cumsum(probs.double(), -1)          # any difference must sit on an fp32-rounding tie of the CDF
    for b in (~same).nonzero().flatten().tolist():
        i = int(got[b])
        assert probs[b, i] > 0 and cdf[b, i] > u[b] - 1e-4 and (i == 0 or cdf[b, i - 1] <= u[b] + 1e-4)
    # RNG path: u from the Philox stream == the explicit-uniform path on the same numbers
    seed, off = 424242, 9
    got_rng = ops.random_sample(probs.to(DEV), seed=seed, offset=off)
    got_u = ops.random_sample(probs.to(DEV), uniform=ops.philox_uniform(B, seed, off))
    assert torch.equal(got_rng, got_u)


@pytest.mark.parametrize("B,V,dtype", [(64, 152064, torch.bfloat16), (16, 152064, torch.float32), (33, 1003, torch.float16),
                                       (5, 8192, torch.bfloat16), (7, 8200, torch.float32)])
def test_softmax_random_sample_equals_softmax_then_random_sample(B, V, dtype):
    """xllm_mi355_softmax_random_sample (round 6) = torch.softmax(logits, -1, fp32) -> random_sample (sampler.cpp:118-137) without
    the [B, V] probabilities: the token equals the oracle's (fp64 CDF of the fp32 softmax) except where u sits within fp32 rounding
    of a CDF step; masked (-inf) and NaN columns are never drawn; rows without mass give 0; do_sample = False rows take the first
    column of the maximum (= argmax of the probabilities)."""
    g = torch.Generator().manual_seed(B * 7 + V)
    logits = (torch.randn(B, V, generator=g) * 3).to(dtype)
    kth = logits.float().topk(min(50, V), -1).values[:, -1:]
    logits[logits.float() < kth] = float("-inf")            # top-k style masking: most columns carry no mass
    logits[1] = float("-inf")                               # nothing valid -> 0
    logits[3, 5] = float("nan")                             # a NaN column has no mass and is not the maximum
    u = torch.rand(B, generator=g)
    u[2] = 1.0                                              # beyond the total -> last column with mass
    do = torch.ones(B, dtype=torch.bool)
    do[4] = False
    clean = torch.where(torch.isnan(logits.float()), torch.tensor(float("-inf")), logits.float())
    probs = torch.softmax(clean, -1)
    probs[1] = 0.0                                          # (softmax of an all -inf row is NaN: no p > 0, random_sample answers 0)
    ref = orc.random_sample(probs, u)
    got = ops.softmax_random_sample(logits.to(DEV), uniform=u.to(DEV), do_sample=do.to(DEV)).cpu()
    assert got[1] == 0
    assert got[4] == int(torch.argmax(clean[4]))            # first column of the maximum
    last_valid = int((probs[2] > 0).nonzero().max())
    assert got[2] == last_valid or probs[2].double().sum() > 1.0 - 1e-6
    cdf = torch.cumsum(probs.double(), -1)
    rows = [b for b in range(B) if b not in (1, 4)]
    same = sum(int(got[b] == ref[b]) for b in rows)
    assert same >= 0.9 * len(rows)
    for b in rows:
        i = int(got[b])
        assert probs[b, i] > 0, (b, i)                      # never a masked / NaN column
        if got[b] != ref[b]:                                # any difference sits on an fp32-rounding tie of the CDF
            assert cdf[b, i] > u[b] - 1e-4 and (i == 0 or cdf[b, i - 1] <= u[b] + 1e-4), (b, i)
    # every row sampled when do_sample is absent; the Philox path equals the explicit-uniform path on the same numbers
    seed, off = 99, 12
    a = ops.softmax_random_sample(logits.to(DEV), seed=seed, offset=off)
    b_ = ops.softmax_random_sample(logits.to(DEV), uniform=ops.philox_uniform(B, seed, off))
    assert torch.equal(a, b_)
    # a strided view (row pitch > V) reads the same rows
    wide = torch.full((B, V + 24), 7.0, dtype=dtype, device=DEV)
    wide[:, :V] = logits.to(DEV)
    assert torch.equal(ops.softmax_random_sample(wide[:, :V], uniform=u.to(DEV), do_sample=do.to(DEV)).cpu(), got)


def test_sample_top_k_top_p_is_the_unfused_sampler_sequence():
    """ops.sample_top_k_top_p (two launches, no [B, V] temporary) against the operator sequence it replaces -- apply_top_k_top_p,
    torch.softmax(fp32), random_sample -- on 16-bit logits at the model's vocabulary: processed logits bit-equal, tokens equal
    except on fp32 CDF ties, and every token is a survivor of the oracle's masking (oracle/sampling.py)."""
    osm = _osm()
    g = torch.Generator().manual_seed(21)
    B, V = 24, 152064
    logits = (torch.randn(B, V, generator=g) * 3).bfloat16()
    temps = torch.rand(B, generator=g) + 0.5
    top_k = torch.randint(1, 200, (B,), generator=g)
    top_k[0] = -1                                           # no limit
    top_p = torch.rand(B, generator=g) * 0.5 + 0.5
    u = torch.rand(B, generator=g)
    for k, p in ((top_k, None), (None, top_p), (top_k, top_p), (None, None)):
        K, P = (None if k is None else k.to(DEV)), (None if p is None else p.to(DEV))
        fused_logits = logits.clone().to(DEV)
        tok = ops.sample_top_k_top_p(fused_logits, temps.to(DEV), K, P, uniform=u.to(DEV)).cpu()
        unf = logits.clone().to(DEV)
        ops.apply_top_k_top_p(unf, temps.to(DEV), K, P)
        assert torch.equal(unf, fused_logits)
        probs = torch.softmax(unf, -1, dtype=torch.float32)
        tok_unf = ops.random_sample(probs, uniform=u.to(DEV)).cpu()
        cdf = torch.cumsum(probs.double().cpu(), -1)
        for b in range(B):
            i = int(tok[b])
            assert bool(torch.isfinite(unf[b, i].float())), (b, i)
            if tok[b] != tok_unf[b]:
                assert cdf[b, i] > u[b] - 1e-4 and (i == 0 or cdf[b, i - 1] <= u[b] + 1e-4), (b, i)
        assert (tok == tok_unf).float().mean() >= 0.9
        if k is not None and p is None:     # the oracle's masking (fp32 restatement on the temperature-scaled 16-bit values)
            base = logits.clone()
            osm.apply_temperatures(base, temps.clone())
            ref = osm.apply_top_k_top_p(base.float(), None, k, None)
            assert bool(torch.isfinite(ref.gather(1, tok.long().view(-1, 1))).all())


def test_random_sample_on_segment_boundaries_stays_next_to_the_crossing():
    """u placed within an ulp of the running sum at the kernel's segment boundaries (multiples of 1024): whichever way the
    two summation orders round, the sampled index has to sit on the crossing of the exact CDF, never far away."""
    V, g = 32768, torch.Generator().manual_seed(77)
    probs = torch.softmax(torch.randn(1, V, generator=g) * 2, -1)
    probs[probs < 1e-6] = 0.0
    cdf = torch.cumsum(probs.double(), -1)[0]
    us, rows = [], []
    for k in range(1, V // 1024):
        c = cdf[k * 1024 - 1].float()
        for uu in (c, torch.nextafter(c, torch.tensor(0.0)), torch.nextafter(c, torch.tensor(2.0))):
            us.append(uu)
    u = torch.stack(us)
    got = ops.random_sample(probs.expand(len(us), V).contiguous().to(DEV), uniform=u.to(DEV)).cpu()
    for b in range(len(us)):
        i = int(got[b])
        assert probs[0, i] > 0 and cdf[i] > u[b] - 1e-5 and (i == 0 or cdf[i - 1] <= u[b] + 1e-5), (b, i, float(u[b]))


def test_moe_index_ignores_ids_outside_the_expert_range():
    """padding rows (-1) or ids >= E are not counted (the reference's histogram guard, moe_compute_index.cu:41-60); their
    src_dst entry is -1 and the sorted combine treats them as absent."""
    T, topk, E, H = 70, 2, 8, 64
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, E, (T, topk), generator=g, dtype=torch.int32)
    ids[3, 1] = -1
    ids[40, 0] = E + 5
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.to(DEV), E)
    flat = ids.flatten()
    ok = (flat >= 0) & (flat < E)
    assert torch.equal(sizes.cpu(), torch.bincount(flat[ok].long(), minlength=E).to(torch.int32))
    sd = src_dst.cpu().flatten()
    assert torch.equal(sd[~ok], torch.full((2,), -1, dtype=torch.int32))
    n_ok = int(ok.sum())
    assert sorted(sd[ok].tolist()) == list(range(n_ok))
    assert torch.equal(dst_src.cpu()[:n_ok][sd[ok].long()], ok.nonzero().flatten().to(torch.int32))
    w = torch.rand(T, topk, generator=g)
    rows = torch.randn(T * topk, H, generator=g).bfloat16()
    got = ops.moe_combine_sorted(rows.to(DEV), src_dst, w.to(DEV), T, topk).cpu()
    gathered = torch.zeros(T * topk, H)
    gathered[ok] = rows.float()[sd[ok].long()]
    ref = (gathered.view(T, topk, H) * w[..., None]).sum(1)
    assert torch.allclose(got.float(), ref, rtol=1e-2, atol=1e-2)


def test_rejection_sample_bit_exact():
    g = torch.Generator().manual_seed(77)
    B, V = 48, 4099
    n = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32)
    cu = torch.cumsum(n, 0).to(torch.int32)
    T = int(cu[-1])
    draft = torch.randint(0, V, (T,), generator=g, dtype=torch.int32)
    draft[3] = V + 5                                  # an out-of-range draft id stops its sequence
    dp = torch.softmax(torch.randn(T, V, generator=g) * 2, -1)
    tp = torch.softmax(torch.randn(T, V, generator=g) * 2, -1)
    dp[:, ::7] = 0.0
    rows = torch.arange(T)
    tp[rows[::3], draft[::3].clamp(max=V - 1).long()] *= 50.0     # force some certain accepts
    ur, up = torch.rand(T, generator=g), torch.rand(T, V, generator=g)
    up[5, :10] = 0.0                                  # FLT_MIN clamp path
    bonus = torch.randint(0, V, (B,), generator=g, dtype=torch.int32)
    ref = orc.rejection_sample(draft, n, cu, dp, tp, bonus, ur, up)
    got = ops.rejection_sample(*[t.to(DEV) for t in (draft, n, cu, dp, tp, bonus, ur, up)]).cpu()
    assert torch.equal(got, ref)
    assert (ref == -1).any() and (ref[cu.long() + torch.arange(B)] >= 0).any()   # both stop and bonus paths exercised


@pytest.mark.parametrize("E,topk,scoring,bias,renorm,dtype", [
    (128, 8, "softmax", False, True, torch.bfloat16), (256, 8, "sigmoid", True, True, torch.float32),
    (64, 6, "softmax", False, False, torch.float32), (60, 4, "sigmoid", False, True, torch.float16),
    (512, 8, "sigmoid", True, False, torch.bfloat16)])
def test_moe_fused_topk(E, topk, scoring, bias, renorm, dtype):
    """N4: cuda::moe_fused_topk; ids bit-exact (incl. lower-index tie-break), weights to fp32 rounding"""
    g = torch.Generator().manual_seed(E + topk)
    T = 300
    x = (torch.randn(T, E, generator=g) * 2).to(dtype)
    x[0] = 0.5                                            # a row of exact ties: experts 0..topk-1 must win in order
    x[1, 7] = x[1, 3]                                     # a pairwise tie
    b = (torch.randn(E, generator=g) * 0.1) if bias else None
    w_ref, id_ref = orc.moe_fused_topk(x, topk, renorm, b, scoring)
    w, ids = ops.moe_fused_topk(x.to(DEV), topk, renorm, b.to(DEV) if bias else None, scoring)
    if not bias:
        assert ids[0].tolist() == list(range(topk))      # all-equal scores: lowest indices in order
    assert torch.equal(ids.cpu(), id_ref)
    torch.testing.assert_close(w.cpu(), w_ref, rtol=3e-6, atol=1e-7)


@pytest.mark.parametrize("E,G,kg,topk,scoring,bias,renorm,scale,dtype", [
    (256, 8, 4, 8, "sigmoid", True, True, 2.5, torch.float32),        # DeepSeek-V3: noaux_tc, 8 groups, 4 kept
    (160, 8, 3, 6, "softmax", False, False, 16.0, torch.bfloat16),    # DeepSeek-V2: group_limited_greedy
    (64, 4, 4, 2, "sigmoid", True, True, 1.0, torch.float16),         # the reference's moe_gate_test shape (all groups kept)
    (512, 64, 5, 8, "softmax", False, True, 1.0, torch.float32),      # the widest supported table, 8 experts per group
    (96, 2, 1, 7, "sigmoid", False, True, 0.5, torch.bfloat16)])
def test_moe_grouped_topk(E, G, kg, topk, scoring, bias, renorm, scale, dtype):
    """a16 / N4: dcu::moe_grouped_topk (topk_gate.cpp:59-125 -> aiter grouped_topk / biased_grouped_topk): ids bit-exact
    vs the restated published algorithm (group limit, tie-breaks), weights to fp32 rounding; moe_active_topk dispatch"""
    g = torch.Generator().manual_seed(E + G + topk)
    T = 333
    x = (torch.randn(T, E, generator=g) * 2).to(dtype)
    x[0] = 0.25                                           # all scores equal: groups 0..kg-1, then the lowest experts
    b = (torch.randn(E, generator=g) * 0.1) if bias else None
    w_ref, id_ref = orc.moe_grouped_topk(x, topk, G, kg, renorm, b, scoring, scale)
    w, ids = ops.moe_active_topk(x.to(DEV), topk, G, kg, renorm, b.to(DEV) if bias else None, scoring, scale)
    assert torch.equal(ids.cpu(), id_ref)
    torch.testing.assert_close(w.cpu(), w_ref, rtol=3e-6, atol=1e-7)
    if not bias:
        assert ids[0].tolist() == list(range(topk))
    per = E // G
    grp = (ids.cpu().long() // per)
    assert all(len(set(r.tolist())) <= kg for r in grp)   # never more than topk_group distinct groups per token
    if renorm:
        torch.testing.assert_close(w.sum(-1).cpu(), torch.full((T,), scale), rtol=1e-5, atol=0)
    with pytest.raises(ops.Mi355Error):
        ops.moe_grouped_topk(x.to(DEV), topk, G, G + 1, renorm, None, scoring)            # topk_group > num_expert_group
    if kg * per > 1:
        with pytest.raises(ops.Mi355Error):
            ops.moe_grouped_topk(x.to(DEV), kg * per + 1, G, kg, renorm, None, scoring)   # more picks than candidates


def test_dual_micro_batch_decoder_equals_single_batch_step():
    """two micro-batches on two streams (attention of one half under the linear layers of the other) produce the bits
    of the single-batch step: per-sequence arithmetic is unchanged"""
    import bench
    from xllm_amd import layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(1024, 3, 16, 4, 128, 2048, 4096, 1e-6, 1e6, 4096)
    B, ctx, bs = 512, 300, 128   # both the whole batch and its halves sit in the same decode-attention plan (4 heads per workgroup)
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=11, n_layers=3)
    md, n_blocks = bench.build_metadata(B, ctx, bs, torch.device(DEV), seed=2)
    g = torch.Generator(device=DEV).manual_seed(7)
    mk = lambda: [KVCache(torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16(),
                          torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16()) for _ in model.layers]
    caches = mk()
    tokens = torch.randint(0, args.vocab_size, (B,), device=DEV, generator=g)
    positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=DEV)
    ref = model.forward(tokens, positions, md, caches).clone()
    dual = layers.DualBatchDecoder(model, md, B)
    for it in range(3):   # repeated runs: stream ordering must hold every time
        out = dual.forward(tokens, positions, caches)
        torch.cuda.synchronize()
        bad = ((out.float() - ref.float()).abs().amax(-1) > 0).nonzero().flatten().tolist()
        assert not bad, (it, len(bad), bad[:6], bad[-3:])
    dual.close()
    for _ in range(6):   # the library keeps 8 per-stream scratch slots: decoders that come and go must give theirs back
        d2 = layers.DualBatchDecoder(model, md, B)
        d2.close()
    d3 = layers.DualBatchDecoder(model, md, B)
    assert torch.equal(d3.forward(tokens, positions, caches), ref)
    d3.close()


def test_deepseek_v2_attention_layer_matches_a_plain_restatement():
    """layers.DeepseekV2Attention (deepseek_v2_attention.cpp:264-317): prefill of ragged sequences, then one decode step on
    top of the cache it wrote, against the reference forward restated with plain fp32 torch ops on the same weights
    (non-paged latent per sequence, softmax(scale q.K^T) K[:, :kv_lora], bottom-right causal)"""
    from xllm_amd import layers
    from xllm_amd.attention import AttentionMetadata, KVCache, build_attention_metadata, build_batch_input
    H, heads, q_lora, kv_lora, nope, rope, v = 1024, 16, 384, 512, 128, 64, 128
    bs, lens = 64, [70, 1, 131]
    gd = torch.Generator(device=DEV).manual_seed(12)
    attn = layers.DeepseekV2Attention(H, heads, q_lora, kv_lora, nope, rope, v, 1e-6, torch.bfloat16, DEV, gd)
    blocks = [[5, 2, 9], [7], [1, 8, 3]]
    cache = KVCache(torch.zeros(11, bs, 1, kv_lora + rope, dtype=torch.bfloat16, device=DEV), None)

    def reference(x, pos, seq_ranges, past):
        """fp32 restatement; `past[b]` = latent rows of sequence b already cached; returns (out, new latents)"""
        f = lambda t: t.float()
        rms = lambda t, w: (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)).bfloat16().float() * f(w)
        lat = (f(x) @ f(attn.kv_a_w).T).bfloat16().float()
        ckv = rms(lat[:, :kv_lora], attn.kv_a_norm_w).bfloat16().float()

        def rope_ds(t):  # t [T, n, rope] in deepseek layout (evens || odds), rotate-half with the same 16-bit roundings
            cs = f(attn.cos_sin[pos])
            c, s_ = cs[:, None, :rope // 2], cs[:, None, rope // 2:]
            a, b = t[..., :rope // 2], t[..., rope // 2:]
            r = lambda u: u.bfloat16().float()
            return torch.cat([r(r(a * c) - r(b * s_)), r(r(b * c) + r(a * s_))], -1)
        kpe = rope_ds(layers.to_deepseek_rope_layout(lat[:, kv_lora:].unsqueeze(1))).squeeze(1)
        latn = torch.cat([ckv, kpe], -1)
        q = (rms((f(x) @ f(attn.q_a_w).T).bfloat16().float(), attn.q_a_norm_w).bfloat16().float() @ f(attn.q_b_w).T).bfloat16().float()
        q = q.view(-1, heads, nope + rope)
        qpe = rope_ds(layers.to_deepseek_rope_layout(q[..., nope:]))
        qabs = torch.einsum("thn,hnk->thk", q[..., :nope], f(attn.w_kc_nk.transpose(1, 2))).bfloat16().float()
        qin = torch.cat([qabs, qpe], -1)
        outs = []
        for b, (a0, a1) in enumerate(seq_ranges):
            K = torch.cat([past[b], latn[a0:a1]], 0) if past[b] is not None else latn[a0:a1]
            sq, sk = a1 - a0, K.size(0)
            sc = torch.einsum("qhd,kd->hqk", qin[a0:a1], K) * attn.scale
            mask = torch.arange(sk, device=DEV)[None, :] <= (torch.arange(sq, device=DEV)[:, None] + sk - sq)
            p = torch.softmax(sc.masked_fill(~mask, float("-inf")), -1)
            outs.append(torch.einsum("hqk,kd->qhd", p, K[:, :kv_lora]))
        o = torch.cat(outs, 0).bfloat16().float()
        o = torch.einsum("thk,hkv->thv", o, f(attn.w_vc_nk.transpose(1, 2))).bfloat16().float().flatten(1, 2)
        return (o @ f(attn.o_w).T), latn

    # prefill
    T = sum(lens)
    x = torch.randn(T, H, device=DEV, generator=gd).bfloat16()
    bi = build_batch_input([0] * 3, lens, blocks, bs)
    md = build_attention_metadata(bi, True, False, DEV)
    pos = bi.positions.to(DEV).long()
    out = attn.forward(pos, x, md, cache)
    cu = [0, 70, 71, 202]
    ref, latn = reference(x, pos, list(zip(cu[:-1], cu[1:])), [None] * 3)
    assert ((out.float() - ref).norm() / ref.norm()).item() <= 8e-3
    rows = cache.k_cache.view(-1, kv_lora + rope)[bi.new_cache_slots.to(DEV).long()].float()
    assert ((rows - latn).abs() <= 2.0 ** -7 * latn.abs() + 1e-3).all()          # the latent rows the layer cached
    # one decode step per sequence on top of that cache
    x2 = torch.randn(3, H, device=DEV, generator=gd).bfloat16()
    bi2 = build_batch_input(lens, [n + 1 for n in lens], blocks, bs)
    md2 = build_attention_metadata(bi2, False, False, DEV)
    pos2 = bi2.positions.to(DEV).long()
    out2 = attn.forward(pos2, x2, md2, cache)
    past = [cache.k_cache.view(-1, kv_lora + rope)[bi.new_cache_slots.to(DEV).long()[a:b]].float() for a, b in zip(cu[:-1], cu[1:])]
    ref2, _ = reference(x2, pos2, [(0, 1), (1, 2), (2, 3)], past)
    assert ((out2.float() - ref2).norm() / ref2.norm()).item() <= 8e-3


def test_deepseek_v2_attention_decode_replays_from_a_hip_graph():
    """MLA decode layer (latent projection, cache write, absorbed query, flash-MLA decode, output projections) captured
    once and replayed on new hidden states / positions / slots == eager, bit for bit"""
    from xllm_amd import layers
    from xllm_amd.attention import AttentionMetadata, KVCache
    H, heads, bs, B = 1024, 16, 64, 6
    gd = torch.Generator(device=DEV).manual_seed(21)
    attn = layers.DeepseekV2Attention(H, heads, 384, 512, 128, 64, 128, 1e-6, torch.bfloat16, DEV, gd)
    cache = KVCache(torch.randn(B * 3 + 1, bs, 1, 576, device=DEV, generator=gd).bfloat16(), None)
    table = torch.randperm(B * 3, device=DEV, generator=gd).to(torch.int32).view(B, 3)
    kv_lens = torch.tensor([130, 64, 1, 99, 191, 65], dtype=torch.int32, device=DEV)
    pos = (kv_lens - 1).long()
    slots = (table.long()[torch.arange(B, device=DEV), pos // bs] * bs + pos % bs).to(torch.int32)
    md = AttentionMetadata(q_cu_seq_lens=torch.arange(B + 1, dtype=torch.int32, device=DEV), kv_cu_seq_lens=None,
                           kv_seq_lens=kv_lens, slot_mapping=slots, block_table=table, max_query_len=1, max_seq_len=192)
    x = torch.randn(B, H, device=DEV, generator=gd).bfloat16()
    attn.forward(pos, x, md, cache)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = attn.forward(pos, x, md, cache)
    x.copy_(torch.randn(B, H, device=DEV, generator=gd).bfloat16())
    kv_lens.copy_(torch.tensor([131, 65, 2, 100, 192, 66], dtype=torch.int32, device=DEV))
    pos.copy_((kv_lens - 1).long())
    slots.copy_((table.long()[torch.arange(B, device=DEV), pos // bs] * bs + pos % bs).to(torch.int32))
    g.replay()
    torch.cuda.synchronize()
    ref_cache = cache.k_cache.clone()
    assert torch.equal(out, attn.forward(pos, x, md, cache)) and torch.equal(cache.k_cache, ref_cache)


# ------------------------------------------------------------------------------------------- N3: step-level harness
@pytest.mark.parametrize("temperature", [0.0, 0.8])
def test_decode_engine_steps_equal_a_hand_driven_loop(temperature):
    """xllm_amd.engine.DecodeEngine (host batch builder -> H2D -> device metadata refresh -> one graph replay -> sampler)
    produces, step after step, the tokens and the KV cache of a loop that rebuilds fresh metadata tensors from the oracle
    every step and calls the model eagerly; graph replay == eager engine."""
    from xllm_amd import engine, layers
    from xllm_amd.attention import AttentionMetadata, KVCache
    args = layers.ModelArgs(512, 2, 8, 2, 64, 1024, 1000, 1e-6, 1e4, 4096)
    bs, B, steps = 16, 5, 6
    lens0 = [33, 16, 1, 47, 20]
    g = torch.Generator().manual_seed(5)
    need = [(L + steps + bs - 1) // bs for L in lens0]
    perm = torch.randperm(sum(need) + 3, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    nb = sum(need) + 3
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=7)
    first = torch.randint(0, args.vocab_size, (B,), generator=g).to(torch.int32)

    def fresh_caches():
        gd = torch.Generator(device=DEV).manual_seed(11)
        return [KVCache(torch.empty(nb, bs, 2, 64, dtype=torch.bfloat16, device=DEV).normal_(generator=gd),
                        torch.empty(nb, bs, 2, 64, dtype=torch.bfloat16, device=DEV).normal_(generator=gd))
                for _ in model.layers]

    runs = []
    for use_graph in (True, False):
        caches = fresh_caches()
        eng = engine.DecodeEngine(model, caches, bs, lens0, blocks, first.to(DEV), max(lens0) + steps, temperature, seed=3,
                                  use_graph=use_graph)
        toks = [eng.step() for _ in range(steps)]
        runs.append((torch.stack(toks), caches))
        assert eng.seq_lens == [L + steps for L in lens0]
    assert torch.equal(runs[0][0], runs[1][0])
    # hand-driven loop: oracle metadata, fresh tensors, eager model
    caches, lens, cur, ref = fresh_caches(), list(lens0), first.clone(), []
    for s in range(steps):
        md_o = orc.build_batch_metadata([n + 1 for n in lens], [1] * B, blocks, bs)
        md = AttentionMetadata(q_cu_seq_lens=md_o["q_cu_seq_lens"].to(DEV), kv_cu_seq_lens=md_o["kv_cu_seq_lens"].to(DEV),
                               kv_seq_lens=md_o["kv_seq_lens"].to(DEV), slot_mapping=md_o["new_cache_slots"].to(DEV),
                               block_table=md_o["block_tables"].to(DEV), max_query_len=1, max_seq_len=max(lens) + 1)
        hidden = model.forward(cur.long().to(DEV), torch.tensor(lens, device=DEV), md, caches)
        logits = model.logits(hidden)
        if temperature <= 0:
            cur = torch.argmax(logits, -1).to(torch.int32).cpu()
        else:
            # the reference's order on the lm_head's own 16-bit logits: div_ in the logits dtype (logits_utils.cpp:54-64), softmax
            # in fp32 (sampler.cpp:118-119), random_sample -- here as the unfused operators, the engine runs the fused launch
            cur = ops.random_sample(torch.softmax(logits.clone().div_(temperature), -1, dtype=torch.float32),
                                    uniform=ops.philox_uniform(B, 3, s, device=DEV)).cpu()
        ref.append(cur)
        lens = [n + 1 for n in lens]
    assert torch.equal(torch.stack(ref), runs[0][0])
    for a, b in zip(caches, runs[0][1]):
        assert torch.equal(a.k_cache, b.k_cache) and torch.equal(a.v_cache, b.v_cache)


# ------------------------------------------------------------------------------------------- weight-stream decode GEMM
_HINT = {"ng": 0, "slices": 0, "rows": 0}


def _hint():
    """xllm_mi355_gemm_plan_hint (include/xllm_mi355.h): the calling thread's planner hint for the packed GEMMs"""
    from xllm_amd import _lib
    _lib.lib().xllm_mi355_gemm_plan_hint(_HINT["ng"], _HINT["slices"], _HINT["rows"])


def _ws_plan(ng, slices):
    _HINT["ng"], _HINT["slices"] = int(ng), int(slices)
    _hint()


@pytest.fixture
def packed_everywhere():
    """ops dispatches to the packed kernel by a measured per-shape policy; these tests want it wherever it is legal"""
    old = ops._PACKED_POLICY
    ops._PACKED_POLICY = "1"
    yield
    ops._PACKED_POLICY = old


def _packed_gemm(a, wp, a_s, w_s, bias, M, N, K, want_acc=False, ws_bytes=64 << 20):
    """straight through the C ABI (no fallback): returns (out, acc or None)"""
    import ctypes as C
    from xllm_amd import _lib
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    acc = torch.empty(M, N, dtype=torch.int32, device=DEV) if want_acc else None
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
    ws.fill_(0x5a)   # the slabs need no zeroing: poison them
    rc = _lib.lib().xllm_mi355_scaled_matmul_packed(
        a.data_ptr(), wp.data_ptr(), a_s.data_ptr(), w_s.data_ptr(), 0 if bias is None else bias.data_ptr(), out.data_ptr(),
        0 if acc is None else acc.data_ptr(), M, N, K, 1, ws.data_ptr() if ws_bytes else 0, ws_bytes,
        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rc, out, acc


def test_pack_weight_i8_layout():
    """packed[((g*K/128 + kt)*2 + ks)*1024 + lane*16 + j] = w[g*16 + (lane & 15)][kt*128 + ks*64 + (lane >> 4)*16 + j]"""
    g = torch.Generator().manual_seed(1)
    N, K = 48, 384
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    wp = ops.pack_weight_i8(w.to(DEV)).cpu().reshape(-1)
    exp = w.view(N // 16, 16, K // 128, 2, 4, 16).permute(0, 2, 3, 4, 1, 5).reshape(-1)   # [g][kt][ks][kq][r][16]
    assert torch.equal(wp, exp)
    assert ops.pack_weight_i8(torch.zeros(40, 384, dtype=torch.int8, device=DEV)) is None       # N % 16
    assert ops.pack_weight_i8(torch.zeros(48, 320, dtype=torch.int8, device=DEV)) is None       # K % 128


def _ws_waves(n):
    """tile height for 128 < M <= 512: 131 = the 256-row eight-wave tile, 128 = 128-row tiles, 0 = planner"""
    _HINT["rows"] = {0: 0, 131: 256, 128: 128}[int(n)]
    _hint()


@pytest.mark.parametrize("M", [129, 200, 256, 257, 512])
def test_packed_gemm_eight_wave_tiles_exact_on_a_few_column_shape(M):
    """round 3 moved 128 < M <= 512 to eight-wave workgroups with the two wave groups one barrier apart (gemm_ws8s_kernel) for
    the wide problems and to 128-row tiles for the few-column ones (N <= 20480: this shape). The 256-row tiles forced onto this
    shape (plan hint tile_rows = 256) must stay exact. (The round-2 four-wave 256-row tile and the in-phase eight-wave arm lost
    their A/Bs and left the library in rounds 3 / 4.)"""
    g = torch.Generator().manual_seed(100 + M)
    N, K = 1936, 1152
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
    wp = ops.pack_weight_i8(w)
    ref_acc = (a.double() @ w.double().T).to(torch.int32)
    ref_out = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None)
    ran = 0
    try:
        _ws_waves(131)
        for ng in (1, 2, 3, 4, 5):
            for slices in (1, 3):
                _ws_plan(ng, slices)
                rc, out, acc = _packed_gemm(a, wp, a_s, w_s, None, M, N, K, want_acc=True)
                assert rc == 0
                ran += 1
                assert torch.equal(acc, ref_acc) and torch.equal(out, ref_out), (ng, slices)
    finally:
        _ws_plan(0, 0)
        _ws_waves(0)
    assert ran == 10


def _packed_gemm_fp8(a, wp, a_s, w_s, bias, M, N, K, ws_bytes=64 << 20):
    from xllm_amd import _lib
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
    ws.fill_(0x7f)   # poisoned slabs (0x7f7f7f7f = 3.4e38 as fp32): nothing may be read before it is written
    rc = _lib.lib().xllm_mi355_fp8_scaled_matmul_packed(
        a.data_ptr(), wp.data_ptr(), a_s.data_ptr(), a_s.numel(), w_s.data_ptr(), w_s.numel(),
        0 if bias is None else bias.data_ptr(), out.data_ptr(), M, N, K, 1, ws.data_ptr() if ws_bytes else 0, ws_bytes,
        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rc, out


@pytest.mark.parametrize("M", [1, 16, 17, 32, 40, 64, 100, 128, 129, 256, 300, 512])
@pytest.mark.parametrize("scales", ["tensor", "tc"])
def test_packed_fp8_gemm_over_tile_shapes(M, scales):
    """fp8 e4m3 weight-stream GEMM on packed weights (a14, round 3): every wave-tile family x width x K slices against the fp64
    product of the dequantised operands. e4m3 x e4m3 products are exact in fp32, so the only freedom is the fp32 summation order:
    |err| <= bf16 rounding + a few accumulation ulps of sum|a||w| -- the bar of test_fp8_gemm_full_size_against_fp64 -- and the
    row-major kernel (another summation order) within the same distance. Ragged N, poisoned slabs, bias."""
    g = torch.Generator().manual_seed(3 * M + len(scales))
    N, K = 1936, 1152
    a = (torch.randn(M, K, generator=g) * 2).to(DEV).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, generator=g) * 0.5).to(DEV).to(torch.float8_e4m3fn)
    per = scales == "tc"
    a_s = (torch.rand(M if per else 1, generator=g) * 0.05 + 0.01).to(DEV)
    w_s = (torch.rand(N if per else 1, generator=g) * 0.02 + 0.01).to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV)
    wp = ops.pack_weight_fp8(w)
    assert wp is not None and wp.dtype == w.dtype
    assert torch.equal(wp.view(torch.int8), ops.pack_weight_i8(w.view(torch.int8)))      # one byte permutation for both kinds
    a64 = a.double() * (a_s.double()[:, None] if per else a_s.double())
    w64 = w.double() * (w_s.double()[:, None] if per else w_s.double())
    ref = a64 @ w64.T + bias.double()
    mag = a64.abs() @ w64.abs().T
    ref_rm = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias)
    ran = 0
    try:
      # (M > 128, N <= 8192 runs on 128-row tiles by default: the second pass forces the 256-row eight-wave tile, arm 131)
      for rows_arm in ((0,) if M <= 128 else (0, 131)):
        _ws_waves(rows_arm)
        for ng in (1, 2, 3, 4, 5, 6, 8, 10):
            for slices in (1, 2, 3):
                _ws_plan(ng, slices)
                rc, out = _packed_gemm_fp8(a, wp, a_s, w_s, bias, M, N, K)
                if rc == -2:
                    continue
                assert rc == 0
                ran += 1
                err = (out.double() - ref).abs()
                assert (err <= 2.0 ** -8 * ref.abs() + 2.0 ** -17 * mag + 1e-6).all(), (ng, slices, float((err / mag).max()))
                assert rel_l2(out, ref) <= 3e-3, (ng, slices)
                assert ((out.double() - ref_rm.double()).abs() <= 2.0 ** -7 * ref.abs() + 2.0 ** -16 * mag + 1e-6).all()
                if slices == 1:   # deterministic: the same launch twice gives the same bits
                    rc2, out2 = _packed_gemm_fp8(a, wp, a_s, w_s, bias, M, N, K)
                    assert rc2 == 0 and torch.equal(out, out2)
    finally:
        _ws_plan(0, 0)
        _ws_waves(0)
    assert ran >= 6
    rc, out = _packed_gemm_fp8(a, wp, a_s, w_s, None, M, N, K, ws_bytes=0)     # no scratch: never sliced, still right
    assert rc == 0 and rel_l2(out, ref - bias.double()) <= 3e-3


def test_packed_fp8_dispatch_and_envelope():
    """ops.fp8_scaled_matmul(b_packed=...) routes decode shapes to the packed kernel and everything else to the row-major one;
    shapes outside the envelope decline with XM_ERR_UNSUPPORTED and have no side effect"""
    g = torch.Generator().manual_seed(5)
    M, N, K = 128, 3072, 1536                     # DeepSeek-V3 q_b_proj of one TP = 8 rank
    a = (torch.randn(M, K, generator=g) * 2).to(DEV).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, generator=g) * 0.5).to(DEV).to(torch.float8_e4m3fn)
    a_s = torch.full((1,), 0.03, device=DEV)
    w_s = torch.full((1,), 0.02, device=DEV)
    wp = ops.pack_weight_fp8(w)
    ref = (a.double() @ w.double().T) * 0.03 * 0.02
    old = ops._PACKED_FP8_POLICY
    try:
        ops._PACKED_FP8_POLICY = "1"
        y1 = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp)
        ops._PACKED_FP8_POLICY = "0"
        y0 = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp)
    finally:
        ops._PACKED_FP8_POLICY = old
    assert rel_l2(y1, ref) <= 3e-3 and rel_l2(y0, ref) <= 3e-3
    rc, _ = _packed_gemm_fp8(a[:, :384].contiguous(), wp, a_s, w_s, None, M, N, 384)        # K < 512
    assert rc == -2
    assert ops.pack_weight_fp8(torch.zeros(40, 384, device=DEV).to(torch.float8_e4m3fn)) is None


@pytest.mark.parametrize("M", [1, 15, 16, 17, 32, 33, 64, 65, 100, 128, 129, 250, 256, 257, 400, 512])
def test_packed_gemm_exact_over_tile_shapes(M):
    """every wave-tile family (M decides it) x every tile width x K slices: exact int32 sums (vs an fp64 product) and an
    output bit-identical to the row-major kernel's; ragged N (a tail tile with fewer live column groups), poisoned slabs"""
    g = torch.Generator().manual_seed(M)
    N, K = 1936, 1152          # 121 column groups (odd: every width leaves a tail), 9 K tiles
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV)
    wp = ops.pack_weight_i8(w)
    ref_acc = (a.double() @ w.double().T).to(torch.int32)
    ref_out = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias)
    ran = 0
    try:
      # (M > 128 with N <= 8192 runs on 128-row tiles by default, m tiles > 1; arm 131 forces the 256-row eight-wave tile here)
      for rows_arm in ((0,) if M <= 128 else (0, 131)):
        _ws_waves(rows_arm)
        for ng in (1, 2, 3, 4, 5, 6, 8, 10):
            for slices in (1, 2, 3):
                _ws_plan(ng, slices)
                rc, out, acc = _packed_gemm(a, wp, a_s, w_s, bias, M, N, K, want_acc=True)
                if rc == -2:
                    continue          # this width does not exist for this tile family
                assert rc == 0
                ran += 1
                assert torch.equal(acc, ref_acc), (rows_arm, ng, slices)
                assert torch.equal(out, ref_out), (rows_arm, ng, slices)
    finally:
        _ws_plan(0, 0)
        _ws_waves(0)
    assert ran >= 6
    rc, out, _ = _packed_gemm(a, wp, a_s, w_s, None, M, N, K, ws_bytes=0)          # no scratch: never sliced, still right
    assert rc == 0 and torch.equal(out, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None))


@pytest.mark.parametrize("M", [32, 64, 128, 256])
def test_packed_gemm_qwen2_7b_shapes_planner(M, packed_everywhere):
    """the four linears of a Qwen2-7B layer through the planner's own choice: exact sums, outputs bit-identical to the
    row-major kernels (ops.scaled_matmul dispatches to the packed kernel when b_packed is given)"""
    g = torch.Generator().manual_seed(7 + M)
    for N, K in ((4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944)):
        a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
        w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
        a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
        w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
        wp = ops.pack_weight_i8(w)
        rc, out, acc = _packed_gemm(a, wp, a_s, w_s, None, M, N, K, want_acc=True)
        assert rc == 0
        assert torch.equal(acc, (a.double() @ w.double().T).to(torch.int32)), (N, K)
        assert torch.equal(out, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None)), (N, K)
        assert torch.equal(out, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp))


@pytest.mark.parametrize("M,N,K,bias", [(256, 3584, 3584, False), (256, 3584, 18944, False), (37, 512, 1024, True),
                                        (300, 1024, 512, True), (64, 3584, 18944, False)])
def test_packed_gemm_add_norm_fusion_equals_separate_ops(M, N, K, bias, packed_everywhere):
    """scaled_matmul_add_rms_norm on packed weights (K slices summed by the consumer) == scaled_matmul ->
    fused_add_rms_norm (-> int8 quant), bit for bit; twice in a row on the same scratch (nothing to re-zero)"""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.02 + 0.001).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.02 + 0.001).to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV) if bias else None
    res0 = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    nw = (torch.rand(N, generator=g) + 0.5).bfloat16().to(DEV)
    wp = ops.pack_weight_i8(w)
    y = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, b)
    res_ref = res0.clone()
    q_ref, s_ref = ops.rms_norm_dynamic_int8_quant(y.clone(), nw, 1e-6, residual=res_ref)
    res_ref2, y2 = res0.clone(), y.clone()
    ops.fused_add_rms_norm(y2, res_ref2, nw, 1e-6)
    for slices in (0, 1, 2, 4):
        try:
            _ws_plan(0, slices)
            for _ in range(2):
                res_a = res0.clone()
                q, qs = ops.scaled_matmul_add_rms_norm(a, w, a_s, w_s, res_a, nw, 1e-6, b, quantize=True, b_packed=wp)
                assert torch.equal(q, q_ref) and torch.equal(qs, s_ref) and torch.equal(res_a, res_ref), slices
            res_b = res0.clone()
            n16 = ops.scaled_matmul_add_rms_norm(a, w, a_s, w_s, res_b, nw, 1e-6, b, quantize=False, b_packed=wp)
            assert torch.equal(n16, y2) and torch.equal(res_b, res_ref2), slices
        finally:
            _ws_plan(0, 0)


# ------------------------------------------------------------------------------------------- greedy sampler
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,V", [(1, 7), (5, 1000), (3, 152064), (256, 4099)])
def test_greedy_argmax_equals_torch(B, V, dtype):
    """Sampler::greedy_sample (sampler.cpp:160-168): first index of the row maximum, ties and NaN as torch.argmax"""
    g = torch.Generator().manual_seed(B * 31 + V)
    x = torch.randn(B, V, generator=g).to(dtype)
    x[0, V // 2] = x[0].max()            # a tie: the lower index wins
    x[0, V // 3] = x[0].max()
    if B > 1:
        x[1, V - 1] = float("nan")       # NaN beats everything
        x[1, V // 4] = float("inf")
    if B > 2:
        x[2] = -float("inf")             # all equal: index 0
    got = ops.greedy_argmax(x.to(DEV))
    assert got.dtype == torch.int64
    assert torch.equal(got.cpu(), torch.argmax(x.float(), dim=-1))
    sl = x.to(DEV)[:, : V - 1] if V > 8 else x.to(DEV)      # a non-contiguous view (odd row pitch: the scalar path)
    assert torch.equal(ops.greedy_argmax(sl).cpu(), torch.argmax(sl.float().cpu(), dim=-1))


@pytest.mark.parametrize("M", [1, 17, 64, 128, 200, 256, 512])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_matmul_argmax_equals_matmul_then_greedy_argmax(M, dtype):
    """lm_head + Sampler::greedy_sample in one pass (xllm_mi355_matmul_argmax_packed, round 4): the token ids are
    greedy_argmax(matmul_packed(...)) of the SAME tile plan bit for bit (same fp32 sums, same 16-bit rounding), on every width
    of the tile family and both tile heights, with ties (first index wins: duplicated weight rows), a NaN row, a bias, ragged N;
    against the oracle's logits wherever its top-2 margin exceeds one 16-bit ulp (round-3 review, next #5); the winning value
    (what tensor-parallel ranks exchange) is the max logit."""
    g = torch.Generator().manual_seed(7 * M + (dtype == torch.float16))
    N, K = 1936, 576
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    w[1500] = w[300]                      # identical columns: a tie wherever they win -> the lower index
    w[1935] = w[7]
    bias = (torch.randn(N, generator=g) * 0.2).to(dtype)
    bias[1500], bias[1935] = bias[300], bias[7]
    if M > 2:
        a[2, 5] = float("nan")            # a NaN row: every logit NaN -> index 0
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    wp = ops.pack_weight_16(wd)
    old = ops._PACKED_16_POLICY
    ran = 0
    try:
        ops._PACKED_16_POLICY = "1"
        for rows in ((0,) if M <= 128 else (0, 131)):
            _ws_waves(rows)
            for b in (None, bd):
                for ng in (0, 1, 2, 3, 4, 5, 6, 8, 10):
                    _ws_plan(ng, 1)                                              # (the fused form never slices K)
                    logits = ops.matmul(ad, wd, b, b_packed=wp)
                    got = ops.matmul_argmax(ad, wp, N, b, want_value=True)
                    if got is None:          # this width is not in the tile family of this M (the caller's two operators serve)
                        assert ng not in (0, 1, 2)
                        continue
                    idx, val = got
                    want = ops.greedy_argmax(logits)
                    assert torch.equal(idx, want), (rows, ng, b is not None)
                    top = logits.float().gather(1, want.view(-1, 1)).view(-1)
                    ok = (val == top) | (torch.isnan(val) & torch.isnan(top))
                    assert bool(ok.all())
                    ran += 1
        _ws_plan(0, 0)
        _ws_waves(0)
        # the planner's own choice, against the oracle where the decision is not within one ulp
        idx = ops.matmul_argmax(ad, wp, N, bd)
        ref = orc.matmul(a, w, bias).float()
        ref_idx = torch.argmax(ref, -1)
        top2 = ref.topk(2, -1).values
        ulp = top2[:, 0].abs() * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
        clear = (top2[:, 0] - top2[:, 1]) > ulp
        clear &= ~torch.isnan(ref).any(-1)
        assert torch.equal(idx.cpu()[clear], ref_idx[clear]) and int(clear.sum()) >= (M * 3) // 5
        if M > 2:
            assert int(idx[2]) == 0
    finally:
        ops._PACKED_16_POLICY = old
        _ws_plan(0, 0)
        _ws_waves(0)
    assert ran >= 6


def test_model_greedy_tokens_equal_argmax_of_logits():
    """Qwen2Model.greedy_tokens (the fused lm_head + argmax the engine and the bench call) == greedy_argmax(logits(hidden)) on the
    model's own lm_head at a decode batch, and the engine's graph replays it"""
    from xllm_amd import layers
    args = layers.ModelArgs(1024, 1, 16, 4, 64, 2048, 32000, 1e-6, 1e6, 4096)
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=3, n_layers=1)
    gd = torch.Generator(device=DEV).manual_seed(5)
    for B in (1, 64, 256):
        hidden = torch.randn(B, 1024, device=DEV, generator=gd).bfloat16()
        _ws_plan(0, 1)
        want = ops.greedy_argmax(model.logits(hidden))
        _ws_plan(0, 0)
        old = ops._GREEDY_FUSION
        try:
            ops._GREEDY_FUSION = "1"
            got = model.greedy_tokens(hidden)
            assert got.dtype == torch.int64 and torch.equal(got, want), B
            ops._GREEDY_FUSION = "0"
            assert torch.equal(model.greedy_tokens(hidden), ops.greedy_argmax(model.logits(hidden)))
            ops._GREEDY_FUSION = "auto"      # one GPU: the two operators (measured tie); tensor parallel: the fused form
            assert torch.equal(model.greedy_tokens(hidden), ops.greedy_argmax(model.logits(hidden)))
        finally:
            ops._GREEDY_FUSION = old


# ------------------------------------------------------------------------------------------- 16-bit weight-stream GEMM, packed
@pytest.mark.parametrize("M", [1, 16, 33, 64, 100, 128, 200, 256, 512])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_packed_16bit_gemm_over_tile_shapes(M, dtype):
    """kernel::matmul on packed 16-bit weights (a15, round 3; cfg2's linears): every wave-tile family x width x K slices against
    the oracle's F::linear (fp32 accumulation in another order: <= 1 ulp of the 16-bit result on almost every element), ragged N,
    poisoned slabs, bias; identical bits run to run"""
    from xllm_amd import _lib
    g = torch.Generator().manual_seed(M * 3 + (dtype == torch.float16))
    N, K = 1936, 576                         # 121 column groups, 9 K tiles of 64 elements
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, generator=g).to(dtype)
    ref = orc.matmul(a, w, bias)
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    wp = ops.pack_weight_16(wd)
    assert wp is not None
    assert torch.equal(wp.view(torch.int8).view(N, 2 * K), ops.pack_weight_i8(wd.view(torch.int8).view(N, 2 * K)))   # bytes
    ran = 0
    try:
      for rows_arm in ((0,) if M <= 128 else (0, 131)):   # (131: the 256-row eight-wave tile also for this small N)
        _ws_waves(rows_arm)
        for ng in (1, 2, 3, 4, 5, 6, 8, 10):
            for slices in (1, 2, 3):
                _ws_plan(ng, slices)
                out = torch.empty(M, N, dtype=dtype, device=DEV)
                ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
                ws.fill_(0x7f)
                rc = _lib.lib().xllm_mi355_matmul_packed(ad.data_ptr(), wp.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K,
                                                         1 if dtype == torch.bfloat16 else 2, ws.data_ptr(), ws.numel(),
                                                         torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if rc == -2:
                    continue
                assert rc == 0
                ran += 1
                assert_ulp_close(out, ref, dtype, ulps=1.0, min_exact=0.95)
    finally:
        _ws_plan(0, 0)
        _ws_waves(0)
    assert ran >= 6
    old = ops._PACKED_16_POLICY
    try:
        ops._PACKED_16_POLICY = "1"
        y1 = ops.matmul(ad, wd, bd, b_packed=wp)
        y2 = ops.matmul(ad, wd, bd, b_packed=wp)
    finally:
        ops._PACKED_16_POLICY = old
    assert torch.equal(y1, y2)
    assert_ulp_close(y1, ref, dtype, ulps=1.0, min_exact=0.95)


# ------------------------------------------------------------------------------------------- gate_up GEMM with SiLU.mul fused
@pytest.mark.parametrize("M,I,K", [(1, 128, 512), (17, 384, 1024), (64, 1280, 512), (128, 18944, 3584), (256, 18944, 3584),
                                   (300, 640, 512), (512, 2432, 1152), (600, 1024, 512), (1500, 2432, 1152), (8192, 4736, 3584)])
def test_gate_up_silu_mul_fusion_equals_separate_ops(M, I, K):
    """scaled_matmul_silu_mul_quant (gate_up GEMM whose epilogue applies SiLU(gate) * up and folds the row maxima, then one
    quantising pass) == scaled_matmul -> act_and_mul_dynamic_int8_quant, bit for bit (q and scale), on the packed kernels
    (M <= 512: every tile family) and on the 256 x 256 8-phase kernel (larger M); the row-amax scratch is zero again afterwards;
    twice in a row on the same scratch"""
    g = torch.Generator().manual_seed(M + I + K)
    N = 2 * I
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.004 + 0.0005).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.004 + 0.0005).to(DEV)
    wp = ops.pack_weight_i8(w)
    gate_up = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None)
    q_ref, s_ref = ops.act_and_mul_dynamic_int8_quant(gate_up, "silu")
    for _ in range(2):
        out = ops.scaled_matmul_silu_mul_quant(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp)
        assert out is not None
        assert torch.equal(out[0], q_ref) and torch.equal(out[1], s_ref)
    assert all(float(v.abs().max()) == 0.0 for v in ops._row_amax.values())
    if M <= 512:      # every width of the tile family, on the planner's tile height and on the 256-row eight-wave tile
        try:
            for waves in (0, 131):
                _ws_waves(waves)
                for ng in (1, 2, 3, 4, 5, 6, 8, 10):
                    _ws_plan(ng, 0)
                    out = ops.scaled_matmul_silu_mul_quant(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp)
                    assert out is not None and torch.equal(out[0], q_ref) and torch.equal(out[1], s_ref), (waves, ng)
        finally:
            _ws_plan(0, 0)
            _ws_waves(0)
    # row-major weights only (no packed copy): the 8-phase kernel serves any M
    out = ops.scaled_matmul_silu_mul_quant(a, w, a_s, w_s, torch.bfloat16, None, b_packed=None)
    assert out is not None and torch.equal(out[0], q_ref) and torch.equal(out[1], s_ref)


@pytest.mark.parametrize("M", [1, 256, 8192])
def test_gate_up_act_row_major_only_straight_through_the_c_abi(M):
    """the round-3 driver-red case, pinned at the C ABI: row-major weights only (w_packed = NULL), M in {1, 256, 8192}. The call
    must either write act + row maxima that are bit-equal to scaled_matmul -> act_and_mul (and non-zero), or decline with
    XM_ERR_UNSUPPORTED -- never return XM_OK with nothing written. (Re-run under every GEMM selector by test_zz_gpu_variants.)"""
    from xllm_amd import _lib
    g = torch.Generator().manual_seed(77 + M)
    I, K = 512, 1024
    N = 2 * I
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    a_s = (torch.rand(M, generator=g) * 0.004 + 0.0005).to(DEV)
    w_s = (torch.rand(N, generator=g) * 0.004 + 0.0005).to(DEV)
    gate_up = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, None)
    ref = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    ops.act_and_mul(ref, gate_up, "silu")
    act = torch.full((M, I), float("nan"), dtype=torch.bfloat16, device=DEV)
    amax = torch.zeros(M, dtype=torch.float32, device=DEV)
    rc = _lib.lib().xllm_mi355_scaled_matmul_gate_up_act(a.data_ptr(), w.data_ptr(), 0, a_s.data_ptr(), w_s.data_ptr(), 0,
                                                         act.data_ptr(), amax.data_ptr(), M, N, K, 1, 0, 0,
                                                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc in (0, -2), rc
    if rc == 0:
        assert float(ref.float().abs().max()) > 0
        assert torch.equal(act, ref)
        assert torch.equal(amax, ref.float().abs().amax(dim=1))
    else:   # a decline leaves the outputs untouched
        assert bool(torch.isnan(act.float()).all()) and float(amax.abs().max()) == 0.0
    # and the operator the layers call: fused result or None, never zeros
    out = ops.scaled_matmul_silu_mul_quant(a, w, a_s, w_s, torch.bfloat16, None, b_packed=None)
    q_ref, s_ref = ops.act_and_mul_dynamic_int8_quant(gate_up, "silu")
    if out is not None:
        assert int(out[0].abs().max()) > 0 and torch.equal(out[0], q_ref) and torch.equal(out[1], s_ref)
    assert rc == 0 or out is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,I,K", [(1, 128, 512), (17, 384, 1024), (64, 18944, 3584), (128, 1280, 512), (300, 640, 512),
                                   (512, 2432, 1152)])
def test_gate_up_silu_mul_fusion_16bit_equals_separate_ops(M, I, K, dtype):
    """matmul_silu_mul (16-bit gate_up projection on packed weights with SiLU(gate) * up in its epilogue) against the packed
    matmul -> act_and_mul: the same expression on fp32 sums whose ORDER depends on the tile plan (a workgroup starts its K walk
    at a tile that is a function of its block index), so >= 98 % of the elements are bit-identical and the rest are one
    or two 16-bit ulps apart (a gate that flips one ulp moves silu * up), every width of the tile family; the torch fp32
    reference to 16-bit rounding; with a bias"""
    g = torch.Generator().manual_seed(3 * M + I + K)
    N = 2 * I
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    wp = ops.pack_weight_16(w)
    old = ops._PACKED_16_POLICY
    try:
        ops._PACKED_16_POLICY = "1"
        for b in (None, bias):
            _ws_plan(0, 1)                                              # the unfused reference without K slices
            gate_up = ops.matmul(a, w, b, b_packed=wp)
            _ws_plan(0, 0)
            ref = torch.empty(M, I, dtype=dtype, device=DEV)
            ops.act_and_mul(ref, gate_up, "silu")
            out = ops.matmul_silu_mul(a, w, b, b_packed=wp)
            assert out is not None
            assert_ulp_close(out, ref, dtype, ulps=2.0, min_exact=0.98)
            for waves in (0, 131):
                _ws_waves(waves)
                for ng in (1, 2, 3, 4, 5, 6, 8, 10):
                    _ws_plan(ng, 0)
                    o2 = ops.matmul_silu_mul(a, w, b, b_packed=wp)   # (None: this width is not in the tile family of this M)
                    if o2 is not None:
                        assert_ulp_close(o2, ref, dtype, ulps=2.0, min_exact=0.98)
            _ws_plan(0, 0)
            _ws_waves(0)
            y = a.float() @ w.float().t() + (b.float() if b is not None else 0.0)
            y = y.to(dtype).float()
            want = (torch.nn.functional.silu(y[:, :I]).to(dtype).float() * y[:, I:]).to(dtype)
            assert_ulp_close(out, want, dtype, ulps=4.0, min_exact=0.75)
    finally:
        ops._PACKED_16_POLICY = old
        _ws_plan(0, 0)
        _ws_waves(0)


@pytest.mark.parametrize("T,d", [(1024, 768), (600, 512), (40, 768), (64, 3584), (24, 24576)])
def test_act_quant_live_sizes_is_the_same_contract_on_every_kernel_path(T, d):
    """act_and_mul_dynamic_int8_quant(live_sizes=...) on the one-wave-per-row kernel (short rows, many tokens), the
    register-resident kernel and the LDS-staged kernel: rows below sum(live_sizes) equal the unmasked operator bit for bit, rows
    past it get scale 0 (never a NaN out of uninitialised memory: the input rows there are poisoned) and are not read"""
    from xllm_amd import _lib
    g = torch.Generator().manual_seed(T + d)
    x = torch.randn(T, 2 * d, generator=g).bfloat16().to(DEV)
    sizes = torch.tensor([T // 4, 0, T // 8, 3], dtype=torch.int32, device=DEV)
    live = int(sizes.sum())
    ref_q, ref_s = ops.act_and_mul_dynamic_int8_quant(x, "silu")
    xp = x.clone()
    xp[live:] = float("nan")                                   # dead input rows must not be read
    q = torch.full((T, d), 77, dtype=torch.int8, device=DEV)
    s = torch.full((T,), float("nan"), dtype=torch.float32, device=DEV)
    rc = _lib.lib().xllm_mi355_act_and_mul_dynamic_int8_quant_live(q.data_ptr(), s.data_ptr(), xp.data_ptr(), T, d, 0, 1,
                                                                  sizes.data_ptr(), sizes.numel(),
                                                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(q[:live], ref_q[:live]) and torch.equal(s[:live], ref_s[:live])
    assert bool((s[live:] == 0).all()) and bool((q[live:] == 77).all())


def test_gate_up_fusion_declines_outside_its_envelope():
    a = torch.zeros(8, 512, dtype=torch.int8, device=DEV)
    w = torch.zeros(2 * 96, 512, dtype=torch.int8, device=DEV)          # I = 96: not a multiple of the 128-column act tile
    s1, s2 = torch.ones(8, device=DEV), torch.ones(192, device=DEV)
    assert ops.scaled_matmul_silu_mul_quant(a, w, s1, s2) is None


# ------------------------------------------------------------------------------------------- N3: logits processors
def _osm():
    from oracle import sampling as osm
    return osm


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_apply_penalties_bit_exact(dtype):
    """apply_frequency_presence_penalties + apply_repetition_penalties (logits_utils.cpp:24-52) through Sampler::forward's order
    (sampler.cpp:35-48): bit-exact against the oracle's torch restatement for fp32 logits (the dtype the reference's own
    gather / scatter_ pipeline accepts); padded id tables (repeated id 0, count 0), one pair absent at a time"""
    osm = _osm()
    g = torch.Generator().manual_seed(41)
    B, V, U = 7, 5000, 37
    logits = (torch.randn(B, V, generator=g) * 3).to(dtype)
    ids = torch.stack([torch.randperm(V, generator=g)[:U] for _ in range(B)])
    cnt = torch.randint(1, 6, (B, U), generator=g, dtype=torch.int32)
    lens = torch.randint(0, U + 1, (B,), generator=g)
    for b in range(B):
        ids[b, lens[b]:] = 0
        cnt[b, lens[b]:] = 0
    freq, pres = torch.rand(B, generator=g) * 2 - 0.5, torch.rand(B, generator=g) * 2 - 0.5
    rep = torch.rand(B, generator=g) * 1.5 + 0.5
    for use_fp, use_rep in ((True, True), (True, False), (False, True)):
        ref = logits.clone().float()
        if dtype == torch.float32:
            if use_fp:
                osm.apply_frequency_presence_penalties(ref, ids, cnt, freq, pres)
            if use_rep:
                osm.apply_repetition_penalties(ref, ids, rep)
        else:
            # 16-bit logits: the same expressions with the in-place operators' cast points written out (sub_ twice, then the
            # where(...) result stored into the 16-bit tensor): fp32 arithmetic, rounded to the logits dtype after each of them
            rt = lambda t: t.to(dtype).float()
            sc = ref.gather(1, ids)
            if use_fp:
                sc = rt(sc - cnt * freq.unsqueeze(1))
                sc = rt(sc - (cnt > 0) * pres.unsqueeze(1))
            if use_rep:
                sc = rt(torch.where(sc < 0, sc * rep.unsqueeze(1), sc / rep.unsqueeze(1)))
            ref.scatter_(1, ids, sc)
        got = logits.clone().to(DEV)
        ops.apply_penalties(got, ids.to(DEV), cnt.to(DEV), freq.to(DEV) if use_fp else None, pres.to(DEV) if use_fp else None,
                            rep.to(DEV) if use_rep else None)
        assert torch.equal(got.float().cpu(), ref), (use_fp, use_rep)
    noop = logits.clone().to(DEV)
    ops.apply_penalties(noop, ids.to(DEV), cnt.to(DEV))
    assert torch.equal(noop.cpu(), logits)


def _mask_mismatch_is_a_boundary_case(ref_row_sorted_probs, p, rank_a, rank_b, exclusive):
    """two top-p cut ranks may differ only when p is within fp32 summation error of the cumulative probability between them"""
    cum = ref_row_sorted_probs.double().cumsum(-1)
    lo, hi = min(rank_a, rank_b), max(rank_a, rank_b)
    pref = cum - ref_row_sorted_probs.double() if exclusive else cum
    return bool(((pref[lo:hi + 1] - p).abs() <= 2e-5).any())


def test_apply_top_k_top_p_degenerate_rows_and_default_k():
    """round-4 advisor: (a) a batch in which ONE request sets top_k carries the others' default top_k = -1: under the both-given
    rule those rows are filtered by p alone (not reduced to top-1); (b) a row without a finite maximum (all -inf) and NaN logits in
    an otherwise finite row do not poison the selection: the all -inf row comes back unchanged, NaN columns carry no mass."""
    from oracle import sampling as osm
    g = torch.Generator().manual_seed(4)
    V = 4096
    base = torch.randn(4, V, generator=g) * 3
    top_k = torch.tensor([50, -1, 0, 7], dtype=torch.int64)
    top_p = torch.tensor([0.9, 0.9, 0.5, 1.0])
    ref = osm.apply_top_k_top_p(base.clone(), None, top_k, top_p)
    got = ops.apply_top_k_top_p(base.clone().to(DEV), None, top_k.to(DEV), top_p.to(DEV)).cpu()
    for b in range(4):       # the same survivors as the sorting reference (one more or less exactly at the cumulative-p boundary)
        kg, kr = ~torch.isinf(got[b]), ~torch.isinf(ref[b])
        assert abs(int(kg.sum()) - int(kr.sum())) <= 1 and int((kg ^ kr).sum()) <= 1, (b, int(kg.sum()), int(kr.sum()))
        assert torch.equal(got[b][kg & kr], ref[b][kg & kr])
    assert int((~torch.isinf(got[1])).sum()) > 50 and int((~torch.isinf(got[2])).sum()) > 1      # k <= 0: NOT greedy
    bad = base.clone()
    bad[1] = float("-inf")
    bad[3, 5] = float("nan")
    out = ops.apply_top_k_top_p(bad.clone().to(DEV), None, top_k.to(DEV), top_p.to(DEV)).cpu()
    assert torch.isinf(out[1]).all() and (out[1] < 0).all()                                         # untouched
    keep3 = ~torch.isinf(out[3])
    assert int(keep3.sum()) <= 8 and torch.equal(out[0], got[0])                                    # k = 7 (+ the NaN column at most)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [1000, 152064])
def test_apply_top_k_top_p_matches_the_sorting_reference(V, dtype):
    """apply_top_k_top_p (logits_utils.cpp:92-155) without a sort: temperature only, top-k only (k <= 0 disables, k >= V keeps all,
    ties at the k-th value by column index), top-p only (exclusive prefix), both (clamped k, inclusive prefix, rank 0 forced) --
    the surviving set equals the oracle's (stable descending torch.sort + softmax + cumsum) except where p sits within fp32
    summation error of a cumulative-probability step; surviving values are bit-equal (temperature division included); identical
    run to run"""
    osm = _osm()
    g = torch.Generator().manual_seed(V + (dtype == torch.bfloat16))
    B = 9
    logits = (torch.randn(B, V, generator=g) * 2.5).to(dtype)
    logits[1, 7] = logits[1, 900] = logits[1, 13] = logits[1].max()            # ties at the top
    logits[2] = logits[2].float().round().to(dtype)                            # many ties everywhere
    temps = torch.tensor([1.0, 0.7, 0.0, 1.3, 2.0, 0.5, 1.0, 0.9, 1.1])
    top_k = torch.tensor([50, 2, 5, 0, V + 5, 1, 300, -1, 17], dtype=torch.int64)
    top_p = torch.tensor([0.9, 0.5, 0.95, 0.0, 1.0, 0.3, 0.999, 0.75, 0.6])
    cases = [("temp", temps, None, None), ("k", None, top_k, None), ("p", None, None, top_p), ("temp+k", temps, top_k, None),
             ("temp+p", temps, None, top_p), ("both", None, top_k, top_p), ("temp+both", temps, top_k, top_p)]
    for name, t, k, p in cases:
        ref = osm.apply_top_k_top_p(logits.clone().float(), None if t is None else t.clone(), k, p)
        base = logits.clone().float()
        if t is not None:
            osm.apply_temperatures(base, t.clone())
        if dtype != torch.float32:
            base = base.to(dtype).float()
            # 16-bit logits: the oracle masks the rows the kernel sees (temperature applied and rounded to the dtype first)
            ref = osm.apply_top_k_top_p(base.clone(), None, k, p)
        got = logits.clone().to(DEV)
        ops.apply_top_k_top_p(got, None if t is None else t.to(DEV), None if k is None else k.to(DEV), None if p is None else p.to(DEV))
        got2 = logits.clone().to(DEV)
        ops.apply_top_k_top_p(got2, None if t is None else t.to(DEV), None if k is None else k.to(DEV), None if p is None else p.to(DEV))
        assert torch.equal(got, got2), name                                      # deterministic
        got = got.float().cpu()
        gm, rm = torch.isinf(got) & (got < 0), torch.isinf(ref) & (ref < 0)
        assert torch.equal(got[~gm], base[~gm]), name                            # survivors keep their (temperature-scaled) value
        for b in range(B):
            if torch.equal(gm[b], rm[b]):
                continue
            assert p is not None, (name, b, int(gm[b].sum()), int(rm[b].sum()))  # top-k alone is exact
            # same top-k set; the top-p cut may move by the ranks whose prefix is within summation error of p
            srt, idx = base[b].sort(dim=-1, descending=True, stable=True)
            if k is not None:
                kk = int(k[b])
                kk = max(1, min(kk, V)) if p is not None and k is not None and name.endswith("both") else (V if kk <= 0 else min(kk, V))
                srt[kk:] = float("-inf")
            probs = srt.softmax(-1)
            n_got, n_ref = int((~gm[b]).sum()), int((~rm[b]).sum())
            assert torch.equal(~gm[b][idx][:n_got], torch.ones(n_got, dtype=torch.bool)), (name, b)   # a PREFIX of the sorted order
            assert _mask_mismatch_is_a_boundary_case(probs, float(p[b]), n_got - 1, n_ref - 1, exclusive=not name.endswith("both")), \
                (name, b, n_got, n_ref)
        assert int((~gm).sum(-1).min()) >= 1, name                               # at least one survivor per row


def test_sampler_pipeline_penalties_temperature_top_k_top_p_then_random_sample():
    """Sampler::forward's non-greedy path (sampler.cpp:33-125) end to end on kernels: penalties -> temperatures + top-k + top-p ->
    softmax (fp32) -> random_sample: the processed logits equal the oracle's and the sampled token is one of the survivors"""
    osm = _osm()
    g = torch.Generator().manual_seed(77)
    B, V, U = 16, 32000, 20
    logits = (torch.randn(B, V, generator=g) * 3)
    ids = torch.randint(0, V, (B, U), generator=g)
    cnt = torch.randint(0, 4, (B, U), generator=g, dtype=torch.int32)
    freq, pres, rep = torch.rand(B, generator=g), torch.rand(B, generator=g), torch.rand(B, generator=g) + 0.8
    temps, top_k = torch.rand(B, generator=g) + 0.5, torch.randint(1, 100, (B,), generator=g)
    ref = logits.clone()
    # (a token id may repeat inside a row of this random table: the oracle's scatter_ order would then be unspecified -- dedupe)
    for b in range(B):
        u, first = torch.unique(ids[b], return_inverse=False, return_counts=False), None
        ids[b, :u.numel()] = u
        ids[b, u.numel():] = u[0]
        cnt[b, u.numel():] = cnt[b, 0]
    osm.apply_frequency_presence_penalties(ref, ids, cnt, freq, pres)
    osm.apply_repetition_penalties(ref, ids, rep)
    ref = osm.apply_top_k_top_p(ref, temps.clone(), top_k, None)
    got = logits.clone().to(DEV)
    ops.apply_penalties(got, ids.to(DEV), cnt.to(DEV), freq.to(DEV), pres.to(DEV), rep.to(DEV))
    ops.apply_top_k_top_p(got, temps.to(DEV), top_k.to(DEV), None)
    assert torch.equal(got.cpu(), ref)
    probs = torch.softmax(got, -1, dtype=torch.float32)
    tok = ops.random_sample(probs, seed=3, offset=0).long().cpu()
    assert bool(torch.isfinite(ref.gather(1, tok.view(-1, 1))).all())
