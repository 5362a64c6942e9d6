"""The C++ libtorch shim (shim/mi355_ops_api.{h,cpp}: xllm::kernel::mi355::* with the reference's exact
torch::Tensor signatures, and shim/mi355_attention.*: the AttentionImpl contract) -- the piece a maintainer
compiles into xLLM with -DUSE_MI355.  CPU: it builds and imports.  GPU: same bits as the ctypes path + oracle."""
import importlib.util
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shim():
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import build_shim
    path = build_shim.main()
    spec = importlib.util.spec_from_file_location("xllm_mi355_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shim_builds_and_exposes_reference_operator_names():
    m = _shim()
    for name in ("rms_norm", "fused_add_rms_norm", "act_and_mul", "reshape_paged_cache", "rotary_embedding", "matmul",
                 "scaled_quantize", "scaled_matmul", "fp8_scaled_quantize", "paged_attention", "attention_forward",
                 "random_sample", "rejection_sample", "moe_fused_topk", "moe_grouped_topk", "moe_active_topk", "moe_gen_idx",
                 "moe_combine_result", "moe_combine_result_sorted", "group_gemm", "group_gemm_gather", "group_gemm_w8a8", "mla_decode",
                 "flash_mla_dense_decode", "flash_mla_prefill_paged", "flash_mla_store_latent_cache", "block_copy", "bmm_heads",
                 "attention_prefill_forward", "piecewise_replay"):
        assert hasattr(m, name)
    from xllm_amd import _lib
    assert m.abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "shim", "mi355_ops_api.h")).read()
    for sym in ("rotary_embedding", "act_and_mul", "reshape_paged_cache", "rms_norm", "fused_add_rms_norm", "matmul",
                "static_scaled_fp8_quant", "fp8_scaled_quantize", "rms_norm_static_fp8_quant",
                "fused_add_rms_norm_static_fp8_quant", "fp8_scaled_matmul", "fused_qk_norm_rope", "scaled_quantize",
                "scaled_matmul", "group_gemm", "build_block_table_from_paged_kv", "random_sample", "rejection_sample",
                "update_llm_decode_metadata", "moe_fused_topk", "moe_grouped_topk", "moe_active_topk", "moe_gen_idx",
                "moe_combine_result", "group_gemm_gather", "mla_decode", "dense_decode", "prefill_paged", "store_latent_cache",
                "block_copy", "bmm_heads"):
        assert sym + "(" in hdr, sym


@pytest.mark.gpu
def test_shim_matches_oracle_and_ctypes_path():
    from oracle import oracle as orc
    from xllm_amd import ops
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(8)
    T, H = 9, 3584
    x = torch.randn(T, H, generator=g).bfloat16()
    w = (torch.rand(H, generator=g) + 0.5).bfloat16()
    out = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
    m.rms_norm(out, x.to(dev), w.to(dev), 1e-6)
    out2 = torch.empty_like(out)
    ops.rms_norm(out2, x.to(dev), w.to(dev), 1e-6)
    assert torch.equal(out, out2)
    q, s = m.scaled_quantize(x.to(dev))
    qr, sr = orc.scaled_quantize(x)
    assert torch.equal(q.cpu(), qr) and torch.equal(s.cpu(), sr)
    wq = torch.randint(-127, 128, (256, H), generator=g, dtype=torch.int8)
    ws = torch.rand(256, generator=g) * 0.02 + 0.01
    m.clear_packed_weight_cache()
    wqd = wq.to(dev)
    y = m.scaled_matmul(q, wqd, s, ws.to(dev), None)
    yr = orc.scaled_matmul(qr, wq, sr, ws)
    assert (y.float().cpu() - yr.float()).abs().max() <= 2.0 ** -7 * yr.float().abs().max()
    # decode-shaped call (M = 9): the shim packed the weight once and ran the weight-stream kernel -- the bits of the
    # row-major kernel and of the Python mirror's packed path; a second call reuses the copy, an in-place update re-packs
    assert m.packed_weight_cache_size() == 1
    assert torch.equal(y, ops.scaled_matmul(q, wqd, s, ws.to(dev), torch.bfloat16, None, b_packed=ops.pack_weight_i8(wqd)))
    assert torch.equal(m.scaled_matmul(q, wqd, s, ws.to(dev), None), y) and m.packed_weight_cache_size() == 1
    wqd.neg_()
    y_neg = m.scaled_matmul(q, wqd, s, ws.to(dev), None)
    assert m.packed_weight_cache_size() == 1 and torch.equal(y_neg.float(), -y.float())
    big = torch.randint(-127, 128, (600, H), generator=g, dtype=torch.int8).to(dev)     # M = 600: the row-major kernels
    sb = torch.rand(600, generator=g).to(dev) * 0.01
    assert torch.equal(m.scaled_matmul(big, wqd, sb, ws.to(dev), None), ops.scaled_matmul(big, wqd, sb, ws.to(dev), torch.bfloat16))
    # ScaledMatmulParams::c (param.h:852-866), alpha = beta = 1: product + 16-bit add, at a decode shape (second pass) and a
    # prefill shape (the 8-phase kernel's epilogue), in place; another alpha / beta is refused, not ignored
    for rows, (aq, asc) in ((9, (q, s)), (600, (big, sb))):
        cc = torch.randn(rows, 256, generator=g).bfloat16().to(dev)
        base = m.scaled_matmul(aq, wqd, asc, ws.to(dev), None)
        want = (base.float() + cc.float()).bfloat16()
        assert torch.equal(m.scaled_matmul_c(aq, wqd, asc, ws.to(dev), None, cc, 1.0, 1.0, None), want)
        inplace = cc.clone()
        assert torch.equal(m.scaled_matmul_c(aq, wqd, asc, ws.to(dev), None, inplace, 1.0, 1.0, inplace), want)
        assert torch.equal(ops.scaled_matmul(aq, wqd, asc, ws.to(dev), torch.bfloat16, None, c=cc), want)
    with pytest.raises(RuntimeError):
        m.scaled_matmul_c(q, wqd, s, ws.to(dev), None, cc[:9].contiguous(), 1.0, 0.5, None)
    # AttentionImpl::forward (decode): KV write + paged attention
    B, nq, nkv, d, bs = 3, 28, 4, 128, 128
    kv_lens = [300, 129, 517]
    pages = [(L + bs - 1) // bs for L in kv_lens]
    perm = torch.randperm(sum(pages) + 2, generator=g).tolist()
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, [1] * B, blocks, bs)
    kc = torch.randn(sum(pages) + 2, bs, nkv, d, generator=g).bfloat16()
    vc = torch.randn(sum(pages) + 2, bs, nkv, d, generator=g).bfloat16()
    qkv = torch.randn(B, (nq + 2 * nkv) * d, generator=g).bfloat16()
    kc_r, vc_r = kc.clone(), vc.clone()
    k3 = qkv[:, nq * d:(nq + nkv) * d].unflatten(-1, (nkv, d))
    v3 = qkv[:, (nq + nkv) * d:].unflatten(-1, (nkv, d))
    orc.reshape_paged_cache(md["new_cache_slots"], k3, v3, kc_r, vc_r)
    ref = orc.paged_attention(qkv[:, :nq * d].unflatten(-1, (nq, d)), kc_r, vc_r, md["q_cu_seq_lens"], md["kv_seq_lens"],
                              md["block_tables"], 1 / math.sqrt(d))
    qd = qkv.to(dev)
    kc_d, vc_d = kc.to(dev), vc.to(dev)
    got = m.attention_forward(qd[:, :nq * d], qd[:, nq * d:(nq + nkv) * d], qd[:, (nq + nkv) * d:], kc_d, vc_d,
                              md["new_cache_slots"].to(dev), md["kv_seq_lens"].to(dev), md["block_tables"].to(dev),
                              nq, nkv, d, max(kv_lens))
    assert torch.equal(kc_d.cpu(), kc_r) and torch.equal(vc_d.cpu(), vc_r)
    err = (got.float().cpu() - ref.float()).norm() / ref.float().norm()
    assert err < 1e-3


@pytest.mark.gpu
def test_shim_random_sample_uses_the_default_generator_stream():
    """random_sample(probs) draws u from the device's default Philox generator exactly like the reference wrapper
    (random_sample.hip:58-74): re-seeding reproduces the tokens, and they equal the explicit (seed, offset) call"""
    from xllm_amd import ops
    m = _shim()
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn(37, 5000, generator=g) * 2, -1).cuda()
    torch.cuda.manual_seed(2024)
    a = m.random_sample(probs)
    b = m.random_sample(probs)                      # the generator offset advanced: a different draw
    torch.cuda.manual_seed(2024)
    assert torch.equal(m.random_sample(probs), a) and not torch.equal(a, b)
    assert torch.equal(a, ops.random_sample(probs, seed=2024, offset=(37 + 3) // 4 * 4))


@pytest.mark.gpu
def test_shim_moe_and_mla_equal_the_ctypes_path():
    """the C++ shim's MoE / MLA operators (what ops_api.cpp's USE_MI355 branch would call) == xllm_amd.ops, bit for bit"""
    from xllm_amd import ops
    m = _shim()
    dev = "cuda"
    gd = torch.Generator(device=dev).manual_seed(4)
    T, E, topk, H, I = 700, 32, 4, 512, 256
    logits = torch.randn(T, E, device=dev, generator=gd).bfloat16()
    w, ids = m.moe_fused_topk(logits, topk, True, None, "softmax")
    w2, ids2 = ops.moe_fused_topk(logits, topk, True)
    assert torch.equal(w, w2) and torch.equal(ids, ids2)
    with pytest.raises(RuntimeError):
        m.moe_fused_topk(logits, topk, True, None, "tanh")
    bias = torch.randn(E, device=dev, generator=gd) * 0.1
    wg, idg = m.moe_active_topk(logits, topk, 8, 3, True, bias, "sigmoid", 2.5)       # grouped (DeepSeek-V3 style)
    wg2, idg2 = ops.moe_grouped_topk(logits, topk, 8, 3, True, bias, "sigmoid", 2.5)
    assert torch.equal(wg, wg2) and torch.equal(idg, idg2)
    wa, ida = m.moe_active_topk(logits, topk, 1, 1, True, None, "softmax", 2.5)        # one group: the plain fused top-k
    assert torch.equal(wa, w) and torch.equal(ida, ids)
    with pytest.raises(RuntimeError):
        m.moe_grouped_topk(logits, topk, 8, 3, True, bias, "softmax", 1.0)             # bias needs sigmoid (topk_gate.cpp:96)
    src_dst, dst_src, sizes = m.moe_gen_idx(ids, E)
    r = ops.moe_compute_index(ids, E)
    assert torch.equal(src_dst, r[0]) and torch.equal(dst_src, r[1]) and torch.equal(sizes, r[2])
    t3 = m.moe_compute_index(ids, E)                      # the cuda_ops_api.h spelling ops_api.cpp:629 calls under the patch
    assert isinstance(t3, tuple) and all(torch.equal(a, b) for a, b in zip(t3, r))
    x = torch.randn(T, H, device=dev, generator=gd).bfloat16()
    w13 = (torch.randn(E, 2 * I, H, device=dev, generator=gd) / 22).bfloat16()
    xs = x.index_select(0, (dst_src // topk).long())
    h = m.group_gemm(xs, w13, sizes)
    assert torch.equal(h, ops.group_gemm(xs, w13, sizes))
    hg = m.group_gemm_gather(x, dst_src, topk, w13, sizes)
    assert hg is not None and torch.equal(hg, h)
    g2 = torch.randn(T * topk, H, device=dev, generator=gd).bfloat16()
    full = torch.empty_like(g2)
    full.index_copy_(0, dst_src.long(), g2)
    out = m.moe_combine_result(full, w)
    assert torch.equal(out, ops.moe_combine_result(full, w, T, topk))
    assert torch.equal(m.moe_combine_result4(full, w, T, topk), out)                  # ops_api.cpp:679 four-argument call
    with pytest.raises(RuntimeError):
        m.moe_combine_result4(full, w, T + 1, topk)
    assert torch.equal(m.moe_combine_result_sorted(g2, w, src_dst), out)
    half = sizes[:E // 2].contiguous()                    # EP form: rows of the other experts count as zero rows
    assert torch.equal(m.moe_combine_result_sorted(g2, w, src_dst, half), ops.moe_combine_sorted(g2, src_dst, w, T, topk, half))
    xq, xs = ops.scaled_quantize(x)
    w13q = torch.randint(-127, 128, (E, 2 * I, H), dtype=torch.int8, device=dev, generator=gd)
    w13s = torch.rand(E, 2 * I, device=dev, generator=gd) * 0.02 + 0.001
    h8 = m.group_gemm_w8a8(xq, xs, w13q, w13s, sizes, dst_src, topk)
    assert torch.equal(h8, ops.group_gemm_w8a8(xq, xs, w13q, w13s, sizes, row_index=dst_src, index_div=topk))
    # MLA decode
    B, Hh, bs = 3, 16, 64
    kv_lens = torch.tensor([200, 64, 1], dtype=torch.int32, device=dev)
    table = torch.randperm(12, device=dev, generator=gd)[:12].to(torch.int32).view(3, 4)
    kc = torch.randn(13, bs, 1, 576, device=dev, generator=gd).bfloat16()
    q = torch.randn(B, Hh, 576, device=dev, generator=gd).bfloat16()
    o = m.mla_decode(q, kc, kv_lens, table, 512, 192 ** -0.5, 200)
    assert torch.equal(o, ops.mla_decode(q, kc, kv_lens, table, 512, 192 ** -0.5, 200))


@pytest.mark.gpu
def test_shim_packed_weight_cache_identity_and_explicit_packing():
    """round-2 advisor (medium): the cache is keyed on the TensorImpl with a weak reference -- a freed weight's entry dies with
    it, a new weight that lands on the recycled address is packed afresh (never fed the old bytes), explicit load-time packing
    registers the copy, and a raw-pointer update is handled by invalidate_packed_weight"""
    from xllm_amd import ops
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(21)
    H = 1024
    a = torch.randint(-127, 128, (8, H), generator=g, dtype=torch.int8).to(dev)
    a_s = torch.rand(8, generator=g).to(dev) * 0.01
    w_s = (torch.rand(256, generator=g) * 0.02 + 0.01).to(dev)
    m.clear_packed_weight_cache()
    results = []
    for i in range(6):      # same shape, freed and re-allocated: torch's caching allocator hands the address out again
        w = torch.randint(-127, 128, (256, H), generator=g, dtype=torch.int8).to(dev)
        ptr = w.data_ptr()
        y = m.scaled_matmul(a, w, a_s, w_s, None)
        assert torch.equal(y, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)), i      # never stale packed bytes
        results.append(ptr)
        del w, y
        assert m.packed_weight_cache_size() == 0             # the entry died with the weight (expired entries are dropped)
    assert len(set(results)) < len(results)                   # (the address really was recycled at least once)
    w = torch.randint(-127, 128, (256, H), generator=g, dtype=torch.int8).to(dev)
    wp = m.pack_w8a8_weight(w)                                # explicit, at weight-load time
    assert torch.equal(wp, ops.pack_weight_i8(w)) and m.packed_weight_cache_size() == 1
    y = m.scaled_matmul(a, w, a_s, w_s, None)
    assert m.packed_weight_cache_size() == 1 and torch.equal(y, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16))
    w.view(torch.uint8).copy_((w.view(torch.uint8) ^ 0x55))   # (torch op: bumps the version -> re-packed by itself)
    assert torch.equal(m.scaled_matmul(a, w, a_s, w_s, None), ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16))
    m.invalidate_packed_weight(w)
    assert m.packed_weight_cache_size() == 0


@pytest.mark.gpu
def test_shim_two_threads_two_streams_share_no_scratch():
    """round-2 review, weak #9 / do-this #6: the reference runs worker THREADS (dist_manager.cpp:82-84); two threads on two
    streams of one GPU hammer the shim's W8A8 linear with decode shapes that slice K (packed kernel: K-slice slabs) and with
    M = 600 (row-major split-K: zero-at-rest scratch). Every result must equal the serial one: the scratch is per
    (device, stream) and every registry is mutex-guarded."""
    import threading
    from xllm_amd import ops
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(33)
    shapes = [(256, 3584, 18944), (64, 3584, 3584), (600, 512, 3584), (256, 4608, 3584)]
    cases = []
    for (M, N, K) in shapes:
        a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
        w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
        a_s = (torch.rand(M, generator=g) * 0.01).to(dev)
        w_s = (torch.rand(N, generator=g) * 0.02 + 0.01).to(dev)
        if M <= 512:
            m.pack_w8a8_weight(w)
        cases.append((a, w, a_s, w_s, ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)))
    torch.cuda.synchronize()
    bad = []

    def worker(tid):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for it in range(40):
                a, w, a_s, w_s, ref = cases[(it + tid) % len(cases)]
                y = m.scaled_matmul(a, w, a_s, w_s, None)
                if it % 8 == 7:
                    st.synchronize()
                if not torch.equal(y, ref):
                    st.synchronize()
                    if not torch.equal(y, ref):
                        bad.append((tid, it))
            st.synchronize()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad


@pytest.mark.gpu
def test_shim_flash_mla_adapter_drives_the_reference_call_sequence():
    """kernel::dcu::flash_mla::dense_decode(DenseDecodeParams&) with the struct of kernels/dcu/flash_mla_adapter.h:33-53, filled the
    way DeepseekV2AttentionImpl::decode_flash_mla fills it (layers/dcu/deepseek_v2_attention.cpp:189-210: separate q_nope / q_pe
    viewed [B, 1, H, .], returns [B, 1, H, kv_lora]), against the ctypes path and the oracle; S_q = 2 (multi-token decode, causal
    and not) against the oracle's bottom-right alignment; the latent store and the paged prefill that replace store_latent_cache /
    prefill_sdpa (:170-178, :212-262); int64 metadata is accepted like the reference's .to(kInt32)."""
    from oracle import oracle as orc
    from xllm_amd import ops
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(19)
    B, H, bs, KVL, ROPE = 5, 16, 64, 512, 64
    kv_lens = [300, 64, 1, 129, 1000]
    pages = [(L + bs - 1) // bs for L in kv_lens]
    nb = sum(pages) + 3
    perm = torch.randperm(nb, generator=g).tolist()
    blocks, used = [], 0
    for n in pages:
        blocks.append(perm[used:used + n]); used += n
    md = orc.build_batch_metadata(kv_lens, [1] * B, blocks, bs)
    kc = torch.randn(nb, bs, 1, KVL + ROPE, generator=g).bfloat16()
    q_nope = torch.randn(B, H, KVL, generator=g).bfloat16()
    q_pe = torch.randn(B, H, ROPE, generator=g).bfloat16()
    scale = 192 ** -0.5
    lens_d, bt_d, kc_d = md["kv_seq_lens"].to(dev), md["block_tables"].to(dev), kc.to(dev)
    out = m.flash_mla_dense_decode(q_nope.to(dev).view(B, 1, H, KVL), q_pe.to(dev).view(B, 1, H, ROPE), kc_d, lens_d, bt_d, KVL,
                                   scale, False)
    assert out.shape == (B, 1, H, KVL)
    q_in = torch.cat([q_nope, q_pe], -1)
    ref = orc.paged_attention(q_in, kc, kc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, dv=KVL)
    assert ((out.view(B, -1).float().cpu() - ref.view(B, -1).float()).norm() / ref.float().norm()).item() <= 1e-3
    # the ctypes path planned with the same upper bound of the lengths gives the same bits
    direct = ops.mla_decode(q_in.to(dev), kc_d, lens_d, bt_d, KVL, scale, bt_d.size(1) * bs)
    assert torch.equal(out.view(B, H, KVL), direct)
    out64 = m.flash_mla_dense_decode(q_nope.to(dev).view(B, 1, H, KVL), q_pe.to(dev).view(B, 1, H, ROPE), kc_d, lens_d.long(),
                                     bt_d.long(), KVL, scale, False)
    assert torch.equal(out64, out)
    # S_q = 2: the last two tokens of every sequence are queries (sequences of length 1 cannot: use lens >= 2)
    lens2 = [max(L, 2) for L in kv_lens]
    md2 = orc.build_batch_metadata(lens2, [2] * B, [b + ([perm[used]] if len(b) * bs < L else []) for b, L in zip(blocks, lens2)], bs)
    q2n = torch.randn(B, 2, H, KVL, generator=g).bfloat16()
    q2p = torch.randn(B, 2, H, ROPE, generator=g).bfloat16()
    for causal in (True, False):
        o2 = m.flash_mla_dense_decode(q2n.to(dev), q2p.to(dev), kc_d, md2["kv_seq_lens"].to(dev), md2["block_tables"].to(dev), KVL,
                                      scale, causal)
        assert o2.shape == (B, 2, H, KVL)
        r2 = orc.paged_attention(torch.cat([q2n, q2p], -1).view(B * 2, H, KVL + ROPE), kc, kc, md2["q_cu_seq_lens"],
                                 md2["kv_seq_lens"], md2["block_tables"], scale, causal=causal, dv=KVL)
        assert ((o2.view(B * 2, -1).float().cpu() - r2.view(B * 2, -1).float()).norm() / r2.float().norm()).item() <= 2.5e-3
    # store_latent_cache + paged prefill (the USE_MI355 branch of DeepseekV2AttentionImpl::forward)
    q_lens = [40, 64, 1, 129, 77]
    md3 = orc.build_batch_metadata(kv_lens, [min(a, b) for a, b in zip(q_lens, kv_lens)], blocks, bs)
    T = int(md3["q_cu_seq_lens"][-1])
    latent = torch.randn(T, KVL + ROPE, generator=g).bfloat16()
    kc_ref = kc.clone()
    orc.reshape_paged_cache(md3["new_cache_slots"], latent.view(T, 1, -1), None, kc_ref, None)
    kc_w = kc.to(dev)
    slots = md3["new_cache_slots"].clone()
    m.flash_mla_store_latent_cache(latent.to(dev), slots.to(dev).long(), kc_w)          # int64 slots, as the reference passes
    assert torch.equal(kc_w.cpu().view(torch.int16), kc_ref.view(torch.int16))
    qn = torch.randn(T, H, KVL, generator=g).bfloat16()
    qp = torch.randn(T, H, ROPE, generator=g).bfloat16()
    o3 = m.flash_mla_prefill_paged(qn.to(dev), qp.to(dev), kc_w, md3["q_cu_seq_lens"].to(dev), md3["kv_seq_lens"].to(dev),
                                   md3["block_tables"].to(dev), KVL, scale, True)
    r3 = orc.paged_attention(torch.cat([qn, qp], -1), kc_ref, kc_ref, md3["q_cu_seq_lens"], md3["kv_seq_lens"], md3["block_tables"],
                             scale, causal=True, dv=KVL)
    assert o3.shape == (T, H, KVL)
    assert ((o3.view(T, -1).float().cpu() - r3.view(T, -1).float()).norm() / r3.float().norm()).item() <= 2.5e-3


@pytest.mark.gpu
def test_shim_block_copy_equals_the_oracle():
    """xllm::kernel::cuda::block_copy as WorkerImpl::execute_cuda_block_copy_kernel calls it (runtime/worker_impl.cpp:1071-1082):
    int64 device arrays of per-layer cache addresses, int32 index tensors, numel_per_block in ELEMENTS + the cache dtype"""
    from oracle import oracle as orc
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    L, nb = 4, 24
    k = [torch.randn(nb, 16, 4, 128, generator=g).bfloat16() for _ in range(L)]
    v = [torch.randn(nb, 16, 4, 128, generator=g).bfloat16() for _ in range(L)]
    kd, vd = [t.to(dev) for t in k], [t.to(dev) for t in v]
    src = torch.tensor([3, 20, 7], dtype=torch.int32)
    dst = torch.tensor([0, 1, 2, 23, 10, 11], dtype=torch.int32)
    cs = torch.tensor([2, 3, 6], dtype=torch.int32)
    orc.block_copy(k, v, src, dst, cs)
    kp = torch.tensor([t.data_ptr() for t in kd], dtype=torch.int64, device=dev)
    vp = torch.tensor([t.data_ptr() for t in vd], dtype=torch.int64, device=dev)
    m.block_copy(kp, vp, src.to(dev), dst.to(dev), cs.to(dev), kd[0][0].numel(), torch.bfloat16)
    torch.cuda.synchronize()
    for l in range(L):
        assert torch.equal(kd[l].cpu(), k[l]) and torch.equal(vd[l].cpu(), v[l])
    with pytest.raises(RuntimeError):                                  # reference CHECK_EQ(src.size(0), cum_sum.size(0))
        m.block_copy(kp, vp, src.to(dev), dst.to(dev), cs[:2].to(dev), kd[0][0].numel(), torch.bfloat16)


@pytest.mark.gpu
def test_shim_attention_prefill_piecewise_capture_and_replay():
    """AttentionImpl::forward on a prefill step the way the DCU graph executor drives it (runtime/dcu_graph_executor_impl.cpp:
    722-800, layers/dcu/flash_attention.cpp:325-365): while capturing, attention is NOT run -- the layer registers a closure and
    returns its pre-allocated output untouched; the replay runs the closure, with the metadata of the step being replayed, into
    that same tensor. Eager result, replay result and the oracle must agree; the KV write happens at capture time either way."""
    from oracle import oracle as orc
    m = _shim()
    dev = "cuda"
    g = torch.Generator().manual_seed(17)
    nq, nkv, d, bs = 8, 2, 128, 16
    lens = [37, 64, 5]
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    q = torch.randn(T, nq * d, generator=g).bfloat16()
    k = torch.randn(T, nkv * d, generator=g).bfloat16()
    v = torch.randn(T, nkv * d, generator=g).bfloat16()
    nb = 16
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int32)
    kc0 = torch.zeros(nb, bs, nkv, d).bfloat16()
    ref = orc.attention_varlen(q.view(T, nq, d), k.view(T, nkv, d), v.view(T, nkv, d), cu, cu, 1.0 / math.sqrt(d), True).view(T, nq, d)
    mk = lambda: (kc0.clone().to(dev), kc0.clone().to(dev))
    # eager
    kc, vc = mk()
    out_e, n = m.attention_prefill_forward(q.to(dev), k.to(dev), v.to(dev), kc, vc, slots.to(dev), cu.to(dev), cu.to(dev),
                                           max(lens), nq, nkv, d, False)
    assert n == 0
    torch.cuda.synchronize()
    assert ((out_e.float().cpu().view(T, nq, d) - ref.float()).norm() / ref.float().norm()) < 2e-3
    kref, vref = kc0.clone(), kc0.clone()
    orc.reshape_paged_cache(slots, k.view(T, nkv, d), v.view(T, nkv, d), kref, vref)
    assert torch.equal(kc.cpu(), kref) and torch.equal(vc.cpu(), vref)
    # capture: one runner registered, the placeholder is returned without attention having run, the KV rows ARE written
    kc2, vc2 = mk()
    qd = q.to(dev)
    out_c, n = m.attention_prefill_forward(qd, k.to(dev), v.to(dev), kc2, vc2, slots.to(dev), cu.to(dev), cu.to(dev),
                                           max(lens), nq, nkv, d, True)
    assert n == 1 and out_c.shape == qd.shape
    out_c.fill_(float("nan"))
    torch.cuda.synchronize()
    assert torch.equal(kc2.cpu(), kref)
    addr = out_c.data_ptr()
    # replay with the captured metadata, then with fresh metadata tensors (what the executor passes at replay)
    m.piecewise_replay(None, None, 0, T)
    torch.cuda.synchronize()
    assert out_c.data_ptr() == addr and torch.equal(out_c, out_e)
    out_c.fill_(float("nan"))
    m.piecewise_replay(cu.clone().to(dev), cu.clone().to(dev), max(lens), T)
    torch.cuda.synchronize()
    assert torch.equal(out_c, out_e)


@pytest.mark.gpu
def test_shim_bmm_heads_equals_the_ctypes_path():
    """xllm::kernel::mi355::bmm_heads as the patched DeepseekV2AttentionImpl calls it (project_output / q_nope absorption)"""
    from xllm_amd import ops
    m = _shim()
    g = torch.Generator().manual_seed(31)
    q = torch.randn(64, 16, 192, generator=g).bfloat16().cuda()
    kv_b = (torch.randn(16, 256, 512, generator=g) / 16).bfloat16().cuda()
    w_kc_nk = kv_b[:, :128].transpose(1, 2).contiguous()      # load_state_dict under USE_MI355
    w_vc_nk = kv_b[:, 128:]                                   # kv_b_proj's own slice, a view
    a = m.bmm_heads(q[..., :128], w_kc_nk)
    assert torch.equal(a, ops.bmm_heads(q[..., :128], w_kc_nk)) and a.shape == (64, 16, 512)
    b = m.bmm_heads(a, w_vc_nk)
    assert torch.equal(b, ops.bmm_heads(a, w_vc_nk)) and b.shape == (64, 16, 128)
    with pytest.raises(RuntimeError):
        m.bmm_heads(q[..., :128], w_vc_nk)                     # K mismatch: the reference's TORCH_CHECK inside bmm


# ------------------------------------------------------------------------------------------- C++ one-shot all-reduce group
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oneshot_msg(rank, i, n, dtype):
    g = torch.Generator().manual_seed(1000 * i + rank)
    return (torch.randn(n, generator=g) * (1 + rank)).to(dtype)


def _oneshot_worker(rank, world, port, ret, fail_rank):
    """one rank of the C++ group (shim/mi355_process_group.cpp) on GPU 0: set-up over a TCPStore as ProcessGroupImpl builds one
    (cuda_process_group.h:49-51), agreed self-test, plain and fused collectives against the single-process operators"""
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = {"ok": True, "log": []}
    try:
        torch.cuda.set_device(0)
        m = _shim()
        from xllm_amd import ops
        fell_back = []

        def fallback(t):            # the group's own all-reduce in an xLLM build (RCCL); here it only records that it was asked
            fell_back.append(t.numel())

        if rank == fail_rank:
            # a rank whose buffer cannot be set up (max_bytes that the allocator refuses) must not hang its peers: everybody gets
            # "inactive" + the same note
            grp = m.OneShotGroup("127.0.0.1", port, rank, world, 0, max_bytes=1 << 50, self_test=True, prefix="tpX", fallback=fallback)
        else:
            grp = m.OneShotGroup("127.0.0.1", port, rank, world, 0, max_bytes=4 << 20, self_test=True, prefix="tpX", fallback=fallback)
        res["active"], res["note"], res["kind"] = grp.active(), grp.note(), grp.allreduce_kind()
        if fail_rank >= 0:
            x = torch.ones(64, dtype=torch.bfloat16, device="cuda")
            grp.allreduce(x)                                  # -> the fallback, on every rank
            res["fell_back"] = list(fell_back)
            ret[rank] = res
            return
        assert grp.active(), grp.note()
        res["grid_limit"], res["memory_kind"] = grp.grid_limit(), grp.memory_kind()
        for i, n in enumerate([8, 3584, 256 * 3584, 24, 1 << 20, 4096]):
            dtype = (torch.bfloat16, torch.float16, torch.float32)[i % 3]
            x = _oneshot_msg(rank, i, n, dtype).cuda()
            want = sum(_oneshot_msg(r, i, n, dtype).float() for r in range(world)).to(dtype)   # fp32 sum in rank order, one rounding
            grp.allreduce(x)
            if not torch.equal(x.cpu(), want):
                res["ok"] = False
                res["log"].append(f"all-reduce mismatch at message {i} (n={n}, {dtype})")
        big = torch.ones(3 << 20, dtype=torch.bfloat16, device="cuda")       # beyond the slot: the backend serves it
        assert not grp.takes(big)
        grp.allreduce(big)
        res["fell_back"] = list(fell_back)
        # fused tail == all-reduce, then the row-wise operator
        for (M, H, quant) in ((256, 3584, True), (70, 7168, False), (1, 128, True)):
            part = (_oneshot_msg(rank, 300 + M, M * H, torch.bfloat16) * 0.5).view(M, H).cuda()
            g0 = torch.Generator().manual_seed(M * 7 + H)
            resid0 = torch.randn(M, H, generator=g0).bfloat16().cuda()
            nw = (torch.rand(H, generator=g0) + 0.5).bfloat16().cuda()
            y = part.clone()
            grp.allreduce(y)
            r_ref = resid0.clone()
            if quant:
                q_ref, s_ref = ops.rms_norm_dynamic_int8_quant(y.clone(), nw, 1e-6, residual=r_ref)
            else:
                n_ref = y.clone()
                ops.fused_add_rms_norm(n_ref, r_ref, nw, 1e-6)
            r_got = resid0.clone()
            got = grp.allreduce_add_rms_norm(part, r_got, nw, 1e-6, quant)
            good = got is not None and torch.equal(r_got, r_ref) and torch.equal(got[2], y)
            good = good and ((torch.equal(got[0], q_ref) and torch.equal(got[1], s_ref)) if quant else torch.equal(got[0], n_ref))
            if not good:
                res["ok"] = False
                res["log"].append(f"fused all-reduce + add + norm mismatch at M={M} H={H} quant={quant}")
        # the GEMM-fed form == packed scaled_matmul -> fused tail, bit for bit (rank 0 carries the bias)
        for (M, N, K, quant) in ((256, 3584, 1792, True), (32, 3584, 896, False)):
            g1 = torch.Generator().manual_seed(1000 * rank + M + K)
            a = torch.randint(-127, 128, (M, K), generator=g1, dtype=torch.int8).cuda()
            w = torch.randint(-127, 128, (N, K), generator=g1, dtype=torch.int8).cuda()
            a_s = (torch.rand(M, generator=g1) * 0.002 + 0.0005).cuda()
            w_s = (torch.rand(N, generator=g1) * 0.002 + 0.0005).cuda()
            g0 = torch.Generator().manual_seed(M + N + K)
            bias = (torch.randn(N, generator=g0).bfloat16() if rank == 0 else torch.zeros(N, dtype=torch.bfloat16)).cuda()
            resid0 = torch.randn(M, N, generator=g0).bfloat16().cuda()
            nw = (torch.rand(N, generator=g0) + 0.5).bfloat16().cuda()
            wp = ops.pack_weight_i8(w)
            part = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, bias, b_packed=wp)
            r_ref = resid0.clone()
            ref = grp.allreduce_add_rms_norm(part, r_ref, nw, 1e-6, quant)
            r_got = resid0.clone()
            got = grp.matmul_allreduce_add_rms_norm(a, a_s, wp, w_s, bias, r_got, nw, 1e-6, quant)
            good = got is not None and ref is not None and torch.equal(r_got, r_ref) and torch.equal(got[0], ref[0])
            good = good and (not quant or torch.equal(got[1], ref[1]))
            if not good:
                res["ok"] = False
                res["log"].append(f"GEMM-fed fused tail mismatch at M={M} N={N} K={K} quant={quant}")
        # a HIP graph holding the collective replays correctly (a plain kernel: no eager piece needed)
        xg = _oneshot_msg(rank, 77, 4096, torch.bfloat16).cuda()
        xin = xg.clone()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                grp.allreduce(xin)
            xin.copy_(xg)
            g.replay()
        torch.cuda.synchronize()
        want = sum(_oneshot_msg(r, 77, 4096, torch.bfloat16).float() for r in range(world)).bfloat16()
        if not torch.equal(xin.cpu(), want):
            res["ok"] = False
            res["log"].append("graph replay of the collective mismatch")
        res["healthy"] = grp.healthy()
        grp.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        res["ok"] = False
        res["log"].append(traceback.format_exc())
    ret[rank] = res


def _run_oneshot_group(fail_rank):
    import torch.multiprocessing as mp
    _shim()                                  # built once, before the ranks race for it
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_oneshot_worker, args=(2, port, ret, fail_rank), nprocs=2, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        assert time.time() - t0 < 240, "the two-rank C++ one-shot group did not finish (a rank hangs in the set-up?)"
    return dict(ret)


@pytest.mark.gpu
def test_cpp_oneshot_group_two_ranks_on_one_gpu():
    """shim/mi355_process_group.cpp (round 6: the C++ host side of the one-shot all-reduce; the reference binds its collectives in
    C++, process_group.cpp:98-110): two processes on one GPU run the store-based set-up, the agreed self-test, plain / fused /
    GEMM-fed collectives bit-equal to the single-process operators, a graph replay, the size fall-back and a collective close()"""
    ret = _run_oneshot_group(fail_rank=-1)
    assert len(ret) == 2
    for r in (0, 1):
        assert ret[r]["ok"], "\n".join(ret[r]["log"])
        assert ret[r]["active"] and ret[r]["note"] == "ok" and ret[r]["kind"] == "oneshot-xgmi"
        assert ret[r]["grid_limit"] == 0                 # two ranks share the GPU: 64 blocks, never 256
        assert ret[r]["fell_back"] == [3 << 20] and ret[r]["healthy"]
    assert ret[0]["memory_kind"] == ret[1]["memory_kind"]


@pytest.mark.gpu
def test_cpp_oneshot_group_declines_together_when_one_rank_cannot_set_up():
    """a rank-local set-up failure travels over the store: BOTH ranks end up inactive with the same note and take the backend
    (nobody hangs in an exchange the other one skipped)"""
    ret = _run_oneshot_group(fail_rank=1)
    assert len(ret) == 2
    for r in (0, 1):
        assert ret[r]["ok"], "\n".join(ret[r]["log"])
        assert not ret[r]["active"] and ret[r]["kind"] == "backend" and ret[r]["fell_back"] == [64]
    assert ret[0]["note"] == ret[1]["note"] and "rank" in ret[0]["note"]
