"""Checks the CPU oracle against the same in-test references the reference's kernel tests use
(tests/core/kernels/dcu/*_test.cpp compare device kernels with torch CPU ops), with the reference's
tolerances as the floor, plus float64 closed forms for the ops the reference never tests
(int8 scaled_quantize / scaled_matmul have no kernel-level reference test; the reference's LAYER fixtures
that run through them are reproduced in tests/test_reference_fixtures.py)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc

torch.manual_seed(2026)
DTYPES = [torch.bfloat16, torch.float16, torch.float32]
TOL = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 1e-2}  # dcu/norm_test.cpp:54-65


def test_seeded_tensor_known_values():
    # FNV-1a("") = 0xcbf29ce484222325; SplitMix64 first output for that seed, computed independently
    import numpy as np
    def ref(key, n):
        h = 0xcbf29ce484222325
        for c in key.encode():
            h = ((h ^ c) * 0x100000001b3) & (2**64 - 1)
        out = []
        st = h
        for _ in range(n):
            st = (st + 0x9E3779B97F4A7C15) & (2**64 - 1)
            z = st
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
            z ^= z >> 31
            out.append(z)
        return out
    u = ref("qwen2_attention_test.k_cache", 7)
    exp = torch.tensor([(x >> 11) / 2.0**53 for x in u], dtype=torch.float64)
    got = orc.seeded_tensor("qwen2_attention_test.k_cache", (7,), torch.float32)
    assert torch.equal(got, exp.to(torch.float32))
    gi = orc.seeded_tensor("qwen2_attention_test.k_cache", (7,), torch.int8)
    assert gi.tolist() == [(-128 + x % 256) for x in u]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conversions_match_torch(dtype):
    x = torch.randn(4096) * 3
    x[:8] = torch.tensor([0.0, -0.0, 65504.0, 65520.0, 1e-8, 6e-8, 3.0e38, -1e-40])
    got = torch.empty(4096, dtype=dtype)
    import ctypes as C
    orc.lib().orc_convert(C.c_void_p(x.data_ptr()), 0, C.c_void_p(got.data_ptr()), orc._DT[dtype], C.c_int64(4096))
    assert torch.equal(got.view(torch.int16) if dtype != torch.float32 else got, x.to(dtype).view(torch.int16) if dtype != torch.float32 else x)


def test_fp8_conversion_matches_torch():
    x = torch.cat([torch.randn(100000) * 50, torch.linspace(-500, 500, 20001), torch.randn(10000) * 0.01])
    x = x.clamp(-448, 448)
    import ctypes as C
    got = torch.empty(x.numel(), dtype=torch.uint8)
    orc.lib().orc_f32_to_e4m3(C.c_void_p(x.data_ptr()), C.c_void_p(got.data_ptr()), C.c_int64(x.numel()))
    ref = x.to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(got, ref)
    back = orc.e4m3_to_f32(torch.arange(256, dtype=torch.uint8))
    refb = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    assert torch.equal(back.nan_to_num(7.0), refb.nan_to_num(7.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H", [(1, 128), (7, 896), (33, 3584)])
def test_rms_norm(dtype, T, H):
    x = torch.randn(T, H).to(dtype)
    w = (torch.rand(H) + 0.5).to(dtype)
    out = torch.empty_like(x)
    orc.rms_norm(out, x, w, 1e-6)
    ref = F.rms_norm(x.float(), (H,), w.float(), 1e-6)  # dcu/norm_test.cpp:113-114 (at::rms_norm on CPU)
    torch.testing.assert_close(out.float(), ref, rtol=TOL[dtype], atol=TOL[dtype])
    # cast-point check (norm.cu:71-72): r16(r16(x*inv)*w)
    inv = torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6).float()
    exact = ((x.float() * inv).to(dtype).float() * w.float()).to(dtype)
    assert (out.float() - exact.float()).abs().max() <= TOL[dtype] * 0.5 * exact.float().abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_add_rms_norm(dtype):
    T, H = 5, 1024
    x = torch.randn(T, H).to(dtype)
    r = torch.randn(T, H).to(dtype)
    w = (torch.rand(H) + 0.5).to(dtype)
    x2, r2 = x.clone(), r.clone()
    orc.fused_add_rms_norm(x2, r2, w, 1e-6)
    z = (x + r)  # 16-bit add (norm.cu:148-150)
    assert torch.equal(r2, z)
    ref = F.rms_norm(z.float(), (H,), w.float(), 1e-6)
    torch.testing.assert_close(x2.float(), ref, rtol=TOL[dtype], atol=TOL[dtype])


def test_rms_norm_fp8_quant():
    T, H = 4, 512
    x = torch.randn(T, H).bfloat16()
    w = (torch.rand(H) + 0.5).bfloat16()
    scale = torch.tensor([0.02])
    out = torch.empty(T, H, dtype=torch.uint8)
    orc.rms_norm_static_fp8_quant(out, x, w, scale, 1e-6)
    inv = torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6).float()
    v = (x.float() * inv).bfloat16().float() * w.float()
    ref = (v * (1.0 / scale)).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    mism = (out != ref).float().mean().item()
    assert mism < 2e-3  # inv computed with a different reduction order may flip a rounding tie


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("neox", [True, False])
def test_rope(dtype, neox):
    T, nq, nk, d = 9, 4, 2, 64
    cache = orc.build_cos_sin_cache(128, d, 10000.0, dtype)
    # cache build vs torch (rotary_embedding_util.cpp:157-192)
    inv_freq = 1.0 / torch.pow(torch.tensor(10000.0), torch.arange(0, d, 2).float() / d)
    fr = torch.outer(torch.arange(128).float(), inv_freq)
    ref_cache = torch.cat([fr.cos(), fr.sin()], -1).to(dtype)
    assert (cache.float() - ref_cache.float()).abs().max() <= (2e-3 if dtype != torch.float32 else 2e-6)
    q = torch.randn(T, nq * d).to(dtype)
    k = torch.randn(T, nk * d).to(dtype)
    pos = torch.randint(0, 128, (T,))
    q2, k2 = q.clone(), k.clone()
    orc.rotary_embedding(pos, q2, k2, cache, d, is_neox=neox)

    def ref(x, n):  # dcu/rope_test.cpp:365-381 style hand RoPE in fp32
        x = x.float().view(T, n, d)
        c = cache[pos, : d // 2].float()[:, None, :]
        s = cache[pos, d // 2:].float()[:, None, :]
        if neox:
            a, b = x[..., : d // 2], x[..., d // 2:]
            return torch.cat([a * c - b * s, b * c + a * s], -1).view(T, n * d)
        a, b = x[..., 0::2], x[..., 1::2]
        return torch.stack([a * c - b * s, b * c + a * s], -1).view(T, n * d)
    tol = {torch.float32: 1e-6, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    torch.testing.assert_close(q2.float(), ref(q, nq), rtol=tol, atol=tol)
    torch.testing.assert_close(k2.float(), ref(k, nk), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["silu", "gelu", "gelu_tanh"])
def test_act_and_mul(dtype, mode):
    T, d = 6, 1000
    x = torch.randn(T, 2 * d).to(dtype)
    out = torch.empty(T, d, dtype=dtype)
    orc.act_and_mul(out, x, mode)
    g, u = x[:, :d].float(), x[:, d:].float()
    act = {"silu": F.silu, "gelu": F.gelu, "gelu_tanh": lambda t: F.gelu(t, approximate="tanh")}[mode]
    ref = act(g) * u  # dcu/activation_test.cpp:84-86
    tol = {torch.float32: 1e-5, torch.float16: 5e-3, torch.bfloat16: 2e-2}[dtype]
    torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol)


def test_reshape_paged_cache_matches_loop():
    # reference_reshape_paged_cache, tests/core/kernels/dcu/reshape_paged_cache_test.cpp:34-50
    T, nkv, d, bs, nb = 37, 4, 128, 16, 12
    qkv = torch.randn(T, (8 + 2 * nkv) * d).bfloat16()
    k = qkv[:, 8 * d: (8 + nkv) * d].unflatten(-1, (nkv, d))  # strided views, as in the layer
    v = qkv[:, (8 + nkv) * d:].unflatten(-1, (nkv, d))
    slots = torch.randperm(nb * bs)[:T].to(torch.int32)
    slots[5] = -1
    kc = torch.zeros(nb, bs, nkv, d, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    orc.reshape_paged_cache(slots, k, v, kc, vc)
    kr, vr = torch.zeros_like(kc), torch.zeros_like(vc)
    for t in range(T):
        s = int(slots[t])
        if s < 0:
            continue
        kr[s // bs, s % bs] = k[t]
        vr[s // bs, s % bs] = v[t]
    assert torch.equal(kc, kr) and torch.equal(vc, vr)


def test_batch_metadata_builder():
    bs = 16
    seq_lens, q_lens = [33, 16, 1, 40], [1, 16, 1, 8]
    blocks = [[5, 0, 9], [7], [3], [2, 11, 4]]
    md = orc.build_batch_metadata(seq_lens, q_lens, blocks, bs)
    assert md["paged_kv_indptr"].tolist() == [0, 3, 4, 5, 8]
    assert md["paged_kv_indices"].tolist() == [5, 0, 9, 7, 3, 2, 11, 4]
    assert md["paged_kv_last_page_len"].tolist() == [1, 16, 1, 8]
    assert md["block_tables"].tolist() == [[5, 0, 9], [7, 0, 0], [3, 0, 0], [2, 11, 4]]  # 0-padded
    assert md["new_cache_slots"][0].item() == 9 * bs + 0  # pos 32 -> block idx 2 (id 9), offset 0
    assert md["new_cache_slots"][1:17].tolist() == [7 * bs + i for i in range(16)]
    assert md["q_cu_seq_lens"].tolist() == [0, 1, 17, 18, 26]
    table = orc.build_block_table_from_paged_kv(md["paged_kv_indptr"], md["paged_kv_indices"])
    assert table.shape == (4, 8) and table[1].tolist() == [7] + [-1] * 7


def test_latent_cache_store_is_index_copy():
    """K-only reshape_paged_cache == k_cache.view(-1, dim).index_copy_(0, slots, latent)
    (DeepseekV2AttentionImpl::store_latent_cache, layers/dcu/deepseek_v2_attention.cpp:170-178)"""
    g = torch.Generator().manual_seed(5)
    kc = torch.randn(6, 64, 1, 576, generator=g).bfloat16()
    latent = torch.randn(50, 576, generator=g).bfloat16()
    slots = torch.randperm(6 * 64, generator=g)[:50].to(torch.int32)
    ref = kc.clone()
    ref.view(-1, 576).index_copy_(0, slots.long(), latent)
    orc.reshape_paged_cache(slots, latent.view(50, 1, 576), None, kc, None)
    assert torch.equal(kc.view(torch.int16), ref.view(torch.int16))


def _decode_metadata_case(seed=3, B=5, B_padded=8, block_size=16):
    """a decode step's staging metadata (BatchInputBuilder rules) + poisoned persistent buffers"""
    import numpy as np
    rng = np.random.default_rng(seed)
    seq_lens = [int(x) for x in rng.integers(1, 70, B)]
    blocks, nxt = [], 1
    for s in seq_lens:
        n = (s + block_size - 1) // block_size
        blocks.append(list(range(nxt, nxt + n)))
        nxt += n
    md = orc.build_batch_metadata(seq_lens, [1] * B, blocks, block_size)
    i32 = lambda x: torch.tensor(np.asarray(x, dtype=np.int32))
    src = dict(tokens=i32(rng.integers(0, 1000, B)), positions=i32([s - 1 for s in seq_lens]),
               new_cache_slots=md["new_cache_slots"], kv_seq_lens=md["kv_cu_seq_lens"],
               paged_kv_indptr=md["paged_kv_indptr"], paged_kv_indices=md["paged_kv_indices"],
               paged_kv_last_page_len=md["paged_kv_last_page_len"])
    max_blocks = max(len(b) for b in blocks) + 2
    poison = lambda *shape: torch.full(shape, -7, dtype=torch.int32)
    dst = dict(tokens=poison(B_padded), positions=poison(B_padded), new_cache_slots=poison(B_padded),
               kv_seq_lens=poison(B_padded + 1), kv_seq_lens_delta=poison(B_padded), paged_kv_indptr=poison(B_padded + 1),
               paged_kv_indices=poison(md["paged_kv_indices"].numel() + 4), paged_kv_last_page_len=poison(B_padded),
               block_table=poison(B_padded, max_blocks), kv_lens=poison(B_padded))
    return src, dst, seq_lens, blocks, md


def test_decode_metadata_update_matches_reference_rules():
    """llm_decode_metadata_update.cu:27-60 restated + the dense block table of batch_input_builder.cpp:904-938"""
    B, Bp = 5, 8
    src, dst, seq_lens, blocks, md = _decode_metadata_case(B=B, B_padded=Bp)
    n_idx = md["paged_kv_indices"].numel()
    orc.decode_metadata_update(src, dst, B, Bp, B, n_idx, Bp)
    assert dst["tokens"].tolist() == src["tokens"].tolist() + [0] * (Bp - B)
    assert dst["new_cache_slots"].tolist() == src["new_cache_slots"].tolist() + [0] * (Bp - B)
    assert dst["positions"].tolist() == src["positions"].tolist() + [-7] * (Bp - B)   # padded tail left alone
    assert dst["kv_seq_lens"][:B + 1].tolist() == md["kv_cu_seq_lens"].tolist()
    assert dst["kv_seq_lens"][B + 1:].tolist() == [-7] * (Bp - B)
    assert dst["kv_seq_lens_delta"][:B].tolist() == seq_lens
    assert dst["paged_kv_indptr"][:B + 1].tolist() == md["paged_kv_indptr"].tolist()
    assert dst["paged_kv_indices"][:n_idx].tolist() == md["paged_kv_indices"].tolist()
    assert dst["paged_kv_indices"][n_idx:].tolist() == [-7] * 4
    assert dst["paged_kv_last_page_len"][:B].tolist() == md["paged_kv_last_page_len"].tolist()
    assert dst["kv_lens"].tolist() == seq_lens + [0] * (Bp - B)
    width = md["block_tables"].size(1)
    assert torch.equal(dst["block_table"][:B, :width], md["block_tables"])          # same 0-padded table
    assert int(dst["block_table"][:, width:].abs().sum()) == 0 and int(dst["block_table"][B:].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_int8_quant_closed_form(dtype):
    M, K = 16, 3584
    x = (torch.randn(M, K) * 2).to(dtype)
    x[3] = 0
    q, s = orc.scaled_quantize(x)
    xd = x.double()
    amax = xd.abs().amax(-1)
    assert torch.allclose(s.double(), amax / 127, rtol=1e-6)
    import numpy as np  # numpy float32 division is correctly rounded (torch computes 127*reciprocal)
    a32 = amax.float().numpy()
    inv = torch.from_numpy(np.where(a32 > 1e-10, np.float32(127.0) / np.maximum(a32, np.float32(1e-30)),
                                    np.float32(0))).unsqueeze(-1)
    ref = torch.round(x.float() * inv).clamp(-127, 127).to(torch.int8)  # torch.round = RNE = nearbyint
    assert torch.equal(q, ref)
    assert q[3].abs().max() == 0 and s[3] == 0


def test_int8_scaled_matmul_closed_form():
    M, N, K = 8, 96, 512
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8)
    a_s = torch.rand(M) * 0.05 + 0.01
    w_s = torch.rand(N) * 0.02 + 0.01
    bias = torch.randn(N).bfloat16()
    out, acc = orc.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias, want_acc=True)
    acc_ref = (a.double() @ w.double().T)
    assert torch.equal(acc.double(), acc_ref)
    ref = (acc_ref.float() * a_s[:, None] * w_s[None, :] + bias.float()).bfloat16()
    assert (out.float() - ref.float()).abs().max() <= 2.0 ** -7 * ref.float().abs().max()
    # w8a8-dynamic linear == matmul with dequantised weight, 5e-2
    # (tests/core/layers/npu_torch/linear_w8a8_dynamic_tests.cpp:110,125-137)
    x = torch.randn(M, K).bfloat16()
    q, s = orc.scaled_quantize(x)
    y = orc.scaled_matmul(q, w, s, w_s, torch.bfloat16)
    ref = x.float() @ (w.float() * w_s[:, None]).T
    assert ((y.float() - ref).abs() / (ref.abs().mean() + ref.abs())).max() < 5e-2


def test_fp8_quant_and_matmul():
    M, N, K = 8, 64, 256
    x = torch.randn(M, K).bfloat16()
    wq = (torch.randn(N, K) * 0.5).to(torch.float8_e4m3fn)
    q, s = orc.fp8_scaled_quantize(x)
    amax = x.abs().max()
    assert torch.equal(s, (amax / 448.0).clamp_min(1e-12).to(torch.float32).view(1))  # fp8_scaled_quantize.cpp:39-44
    ref_q = (x.float() * (1.0 / s)).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q, ref_q)
    w_s = torch.tensor([0.03])
    out = orc.fp8_scaled_matmul(q, wq.view(torch.uint8), s, w_s, torch.bfloat16)
    ref = (q.view(torch.float8_e4m3fn).float() @ wq.float().T) * s * w_s
    # cuda/cutlass_scaled_mm_test.cpp:93,147 uses max_abs<2.0/mean<0.5; we hold 1 bf16 ulp
    assert (out.float() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention_vs_sdpa(dtype):
    nq, nkv, d = 8, 2, 64
    lens = [5, 17, 32]
    cu = torch.tensor([0, 5, 22, 54], dtype=torch.int32)
    T = 54
    q = torch.randn(T, nq, d).to(dtype)
    k = torch.randn(T, nkv, d).to(dtype)
    v = torch.randn(T, nkv, d).to(dtype)
    scale = 1 / math.sqrt(d)
    out = orc.attention_varlen(q, k, v, cu, cu, scale, causal=True).view(T, nq, d)
    for b, L in enumerate(lens):
        s = int(cu[b])
        qq = q[s:s + L].float().permute(1, 0, 2)[None]
        kk = k[s:s + L].float().repeat_interleave(nq // nkv, 1).permute(1, 0, 2)[None]
        vv = v[s:s + L].float().repeat_interleave(nq // nkv, 1).permute(1, 0, 2)[None]
        ref = F.scaled_dot_product_attention(qq, kk, vv, is_causal=True)[0].permute(1, 0, 2)
        tol = 1e-5 if dtype == torch.float32 else 1e-2
        torch.testing.assert_close(out[s:s + L].float(), ref, rtol=tol, atol=tol)


def test_paged_attention_modes_agree():
    """decode and chunked prefill over pages == dense attention on the gathered sequence,
    with BOTTOM-RIGHT causal alignment for chunks (SURVEY 8c caveat 2)."""
    nq, nkv, d, bs, nb = 4, 2, 32, 8, 40
    kv_lens = [19, 8, 33]
    q_lens = [3, 8, 1]
    perm = torch.randperm(nb).tolist()
    blocks, used = [], 0
    for L in kv_lens:
        n = (L + bs - 1) // bs
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, q_lens, blocks, bs)
    kc = torch.randn(nb, bs, nkv, d).bfloat16()
    vc = torch.randn(nb, bs, nkv, d).bfloat16()
    Tq = sum(q_lens)
    q = torch.randn(Tq, nq, d).bfloat16()
    scale = d ** -0.5
    out = orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                              causal=True).view(Tq, nq, d)
    qs = 0
    for b, (L, ql) in enumerate(zip(kv_lens, q_lens)):
        kk = torch.cat([kc[p] for p in blocks[b]], 0)[:L].float().repeat_interleave(nq // nkv, 1)
        vv = torch.cat([vc[p] for p in blocks[b]], 0)[:L].float().repeat_interleave(nq // nkv, 1)
        qq = q[qs:qs + ql].float()
        sc = torch.einsum("qhd,khd->hqk", qq, kk) * scale
        mask = torch.arange(L)[None, :] > (L - ql + torch.arange(ql))[:, None]
        sc = sc.masked_fill(mask[None], float("-inf"))
        ref = torch.einsum("hqk,khd->qhd", sc.softmax(-1), vv)
        torch.testing.assert_close(out[qs:qs + ql].float(), ref, rtol=1e-2, atol=1e-2)
        qs += ql
    with pytest.raises(ValueError):
        bad = md["block_tables"].clone(); bad[0, 0] = nb
        orc.paged_attention(q, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], bad, scale)


def test_moe_helpers():
    T, topk, E, Hd = 13, 2, 4, 32
    ids = torch.randint(0, E, (T, topk), dtype=torch.int32)
    src_dst, dst_src, sizes = orc.moe_compute_index(ids, E)
    assert sizes.tolist() == torch.bincount(ids.flatten().long(), minlength=E).tolist()
    assert sorted(src_dst.tolist()) == list(range(T * topk))
    assert torch.equal(dst_src[src_dst.long()], torch.arange(T * topk, dtype=torch.int32))
    sorted_e = ids.flatten()[dst_src.long()]
    assert torch.all(sorted_e[1:] >= sorted_e[:-1])
    g = torch.randn(T * topk, Hd).bfloat16()
    w = torch.rand(T, topk)
    out = orc.moe_combine(g, w, T, topk)
    ref = (g.float().view(T, topk, Hd) * w[..., None]).sum(1)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("E,G,kg,topk,scoring,bias,renorm,scale", [
    (256, 8, 4, 8, "sigmoid", True, True, 2.5), (160, 8, 3, 6, "softmax", False, False, 16.0),
    (64, 4, 4, 2, "sigmoid", True, True, 1.0), (64, 4, 2, 4, "softmax", False, True, 1.0)])
def test_grouped_topk_oracle_vs_tensorwise_published_gate(E, G, kg, topk, scoring, bias, renorm, scale):
    """the C restatement of the grouped gate (oracle: orc_moe_grouped_topk) against the SAME published algorithm written
    the way the open DeepSeek-V2 / V3 gates write it (view -> group max | top-2 sum -> topk groups -> mask -> topk)"""
    g = torch.Generator().manual_seed(E * 3 + topk)
    T = 200
    x = torch.randn(T, E, generator=g) * 2
    b = torch.randn(E, generator=g) * 0.1 if bias else None
    w, ids = orc.moe_grouped_topk(x, topk, G, kg, renorm, b, scoring, scale)
    s = torch.sigmoid(x) if scoring == "sigmoid" else torch.softmax(x, -1)
    c = s + b if bias else s
    gv = c.view(T, G, -1).topk(2, -1)[0].sum(-1) if bias else c.view(T, G, -1).max(-1)[0]
    keep = torch.zeros(T, G).scatter_(1, gv.topk(kg, -1)[1], 1.0).bool()
    cm = c.masked_fill(~keep[:, :, None].expand(T, G, E // G).reshape(T, E), float("-inf"))
    ref_ids = cm.topk(topk, -1)[1]
    ref_w = s.gather(1, ref_ids)
    if renorm:
        ref_w = ref_w / ref_w.sum(-1, keepdim=True)
    ref_w = ref_w * scale
    assert torch.equal(ids.long(), ref_ids)               # random floats: no ties, so torch.topk's order is the same
    torch.testing.assert_close(w, ref_w, rtol=2e-5, atol=1e-7)
    # tie-breaks and the group limit on a constructed row: all scores equal -> groups 0..kg-1, lowest experts first
    w0, i0 = orc.moe_grouped_topk(torch.full((1, E), 0.25), topk, G, kg, renorm, None, scoring, scale)
    assert i0[0].tolist() == list(range(topk))


def test_grouped_topk_oracle_invariants_over_random_geometries():
    """size-independent properties of the grouped gate over random (E, groups, kept groups, topk) geometries: every pick
    lies in one of at most topk_group groups, picks are distinct and ordered by choice score, weights are the unbiased
    scores (renormalised sum = routed_scaling_factor), and a one-group problem degenerates to the plain top-k"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(2, 16), st.integers(1, 16), st.integers(1, 8), st.booleans(), st.booleans(), st.integers(0, 10 ** 6))
    def run(G, per, topk, sigmoid, renorm, seed):
        E = G * per
        g = torch.Generator().manual_seed(seed)
        kg = int(torch.randint(1, G + 1, (1,), generator=g))
        topk_ = min(topk, kg * per)
        x = torch.randn(9, E, generator=g) * 3
        bias = torch.randn(E, generator=g) * 0.2 if (sigmoid and per >= 2 and seed % 2) else None
        w, ids = orc.moe_grouped_topk(x, topk_, G, kg, renorm, bias, "sigmoid" if sigmoid else "softmax", 1.75)
        s = torch.sigmoid(x) if sigmoid else torch.softmax(x, -1)
        c = s + bias if bias is not None else s
        for t in range(x.size(0)):
            row = ids[t].long()
            assert len(set(row.tolist())) == topk_
            assert len(set((row // per).tolist())) <= kg
            cs = c[t][row]
            assert torch.all(cs[:-1] >= cs[1:])                       # descending choice score
            ref_w = s[t][row]
            ref_w = ref_w / ref_w.sum() * 1.75 if renorm else ref_w * 1.75
            torch.testing.assert_close(w[t], ref_w, rtol=2e-5, atol=1e-7)
        if kg == G and bias is None:                                    # every group kept: the plain top-k over all experts
            assert torch.equal(ids.long(), c.topk(topk_, -1)[1])
    run()


# ------------------------------------------------------------------------------------------- N3 sampler
def test_philox4x32_10_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)"""
    import ctypes as C
    import numpy as np
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_goldens.json")))
    kat = [(tuple(v["ctr"]), tuple(v["key"]), tuple(v["out"])) for v in gold["philox4x32_10_kat"]["vectors"]]
    assert kat[0][2] == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8) and len(kat) == 3
    for ctr, key, want in kat:
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        orc.lib().orc_philox4x32_10(c, k, o)
        assert tuple(o) == want
    # hiprand_init(seed, subsequence, offset) + hiprand_uniform: counter (offset/4, subsequence), word offset%4
    u = orc.philox_uniform(1, 0, 2)       # seed 0, row 0, offset 2 -> third word of the all-zero vector
    assert u.item() == np.float32(2.3283064e-10) + np.float32(0xbc57ac4c) * np.float32(2.3283064e-10)
    assert 0.0 < float(orc.philox_uniform(4096, 1234, 7).min()) and float(orc.philox_uniform(4096, 1234, 7).max()) <= 1.0


def test_random_sample_oracle_rules():
    probs = torch.tensor([[0.1, 0.0, 0.2, 0.7], [0.0, 0.0, 0.0, 0.0], [0.5, 0.5, 0.0, 0.0], [0.25, 0.25, 0.25, 0.25]])
    u = torch.tensor([0.25, 0.5, 1.0, 0.75])
    # row0: cdf .1,.1,.3,1 -> first > .25 is index 2; row1: nothing valid -> 0; row2: total 1.0 !> 1.0 -> last valid 1;
    # row3: cdf .25,.5,.75,1 -> first > .75 is index 3
    assert orc.random_sample(probs, u).tolist() == [2, 0, 1, 3]


def test_rejection_sample_oracle_rules():
    V = 5
    draft = torch.tensor([1, 2, 3, 0, 7], dtype=torch.int32)            # seq0: 2 drafts, seq1: 2 drafts, seq2: 1 (bad id)
    n = torch.tensor([2, 2, 1], dtype=torch.int32)
    cu = torch.tensor([2, 4, 5], dtype=torch.int32)
    dp = torch.full((5, V), 0.2)
    tp = torch.full((5, V), 0.2)
    tp[1] = torch.tensor([0.0, 0.1, 0.0, 0.6, 0.3])                      # seq0 draft 1 (token 2): target 0 -> reject
    ur = torch.tensor([0.5, 0.5, 0.5, 0.5, 0.5])
    up = torch.full((5, V), 0.5)
    bonus = torch.tensor([11, 12, 13], dtype=torch.int32)
    out = orc.rejection_sample(draft, n, cu, dp, tp, bonus, ur, up)
    # seq0: accept token 1 (ratio 1), reject token 2 -> recovered argmax(max(tp-dp,0)/u) = index 3; no bonus
    # seq1: both accepted -> bonus 12; seq2: draft id 7 out of range -> all -1
    assert out.tolist() == [1, 3, -1, 3, 0, 12, -1, -1]


def test_attention_flash_cast_point_mode():
    """oracle p_round="flash" (round 3): the un-normalised 64-key-tile P rounded to the tensor dtype, fp32 row sum, lazy running
    maximum -- a restatement of WHERE a flash kernel rounds P. It must (a) equal the fp32-P result when P needs no rounding
    (fp32 tensors), (b) sit at the distance of ONE 16-bit rounding of P from it for bf16 (the same order as the normalised-P
    spec, p_round=True), (c) not depend on how the keys fall into tiles beyond that distance, (d) honour masks / ragged batches."""
    g = torch.Generator().manual_seed(5)
    nq, nkv, d = 4, 2, 64
    lens = [1, 63, 64, 65, 300]
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    scale = d ** -0.5
    r = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    q32, k32, v32 = (torch.randn(T, n, d, generator=g) for n in (nq, nkv, nkv))
    a0 = orc.attention_varlen(q32, k32, v32, cu, cu, scale)
    a2 = orc.attention_varlen(q32, k32, v32, cu, cu, scale, p_round="flash")
    assert r(a2, a0) < 5e-6                                     # (a) fp32: only summation order / exp2 differ
    q, k, v = q32.bfloat16(), k32.bfloat16(), v32.bfloat16()
    b0 = orc.attention_varlen(q, k, v, cu, cu, scale)
    b1 = orc.attention_varlen(q, k, v, cu, cu, scale, p_round=True)
    b2 = orc.attention_varlen(q, k, v, cu, cu, scale, p_round="flash")
    assert 2e-4 < r(b2, b0) < 6e-3 and 2e-4 < r(b1, b0) < 6e-3  # (b) one bf16 rounding of P each
    assert r(b2, b1) < 8e-3
    # (d) a window and a non-causal call go through the same code path
    w0 = orc.attention_varlen(q, k, v, cu, cu, scale, window_left=70)
    w2 = orc.attention_varlen(q, k, v, cu, cu, scale, window_left=70, p_round="flash")
    assert r(w2, w0) < 6e-3
    n0 = orc.attention_varlen(q32, k32, v32, cu, cu, scale, causal=False)
    n2 = orc.attention_varlen(q32, k32, v32, cu, cu, scale, causal=False, p_round="flash")
    assert r(n2, n0) < 5e-6


# ------------------------------------------------------------------------------------------- KV block copy
def test_block_copy_oracle_hand_computed_groups():
    """cuda::block_copy (kernels/cuda/block_copy.cu:38-118): cum_sum is the inclusive running count of destinations per source.
    Sources [5, 1] with cum_sum [3, 4]: destinations 0..2 copy block 5, destination 3 copies block 1 -- worked by hand."""
    L, nb = 3, 8
    k = [torch.arange(nb * 4, dtype=torch.float32).reshape(nb, 2, 2) + 100 * l for l in range(L)]
    v = [-(torch.arange(nb * 4, dtype=torch.float32).reshape(nb, 2, 2)) - 100 * l for l in range(L)]
    k0, v0 = [t.clone() for t in k], [t.clone() for t in v]
    src = torch.tensor([5, 1], dtype=torch.int32)
    dst = torch.tensor([0, 7, 2, 3], dtype=torch.int32)
    cs = torch.tensor([3, 4], dtype=torch.int32)
    orc.block_copy(k, v, src, dst, cs)
    for l in range(L):
        for d, sblk in ((0, 5), (7, 5), (2, 5), (3, 1)):
            assert torch.equal(k[l][d], k0[l][sblk]) and torch.equal(v[l][d], v0[l][sblk])
        for untouched in (1, 4, 5, 6):
            assert torch.equal(k[l][untouched], k0[l][untouched]) and torch.equal(v[l][untouched], v0[l][untouched])
    # a destination index past the last running count falls into the last group (the kernel's search stops at num_groups - 1)
    k2 = [t.clone() for t in k0]
    orc.block_copy(k2, None, torch.tensor([6], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32),
                   torch.tensor([1], dtype=torch.int32))
    assert torch.equal(k2[0][0], k0[0][6]) and torch.equal(k2[0][1], k0[0][6])
    # no sources: nothing happens (block_copy.cu:128-130)
    k3 = [t.clone() for t in k0]
    orc.block_copy(k3, None, torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32))
    assert all(torch.equal(a, b) for a, b in zip(k3, k0))
