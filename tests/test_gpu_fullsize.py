"""GPU parity at BASELINE.json's FULL sizes (cfg3: Qwen2-7B, batch 256, ctx 4096; prefill chunks of 4096 tokens).

The oracle cannot finish these shapes in seconds, so every test combines
  * an oracle check on a SAMPLE of the units (sequences / rows / queries) of the full-size launch, and
  * size-independent properties of the whole output: physical-page permutation invariance, independence of the
    sequences of a batch, causality (future keys cannot change past outputs), exact integer checksums of the int32
    accumulators (a checksum of checksums), scatter -> gather round trips.
Bars as in test_gpu_parity.py (bit-exact for integer / index work, <= 1e-3 relative L2 for bf16 attention).
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
NQ, NKV, D, H, I = 28, 4, 128, 3584, 18944
B, CTX, BS = 256, 4096, 128


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def _paged_cache(gen, n_seqs, ctx, extra=7):
    pages = (ctx + BS - 1) // BS
    n_blocks = n_seqs * pages + extra
    perm = torch.randperm(n_blocks, generator=gen)[: n_seqs * pages].to(torch.int32).view(n_seqs, pages)
    return n_blocks, perm


def test_paged_decode_full_size_sample_and_properties():
    g = torch.Generator().manual_seed(100)
    gd = torch.Generator(device=DEV).manual_seed(100)
    n_blocks, table = _paged_cache(g, B, CTX)
    kc = torch.empty(n_blocks, BS, NKV, D, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    vc = torch.empty(n_blocks, BS, NKV, D, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    q = torch.empty(B, NQ, D, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    kv_lens = torch.full((B,), CTX, dtype=torch.int32)
    kv_lens[5], kv_lens[77], kv_lens[200] = CTX - 1, 1, 2049          # ragged tails inside the full-size batch
    scale = 1.0 / math.sqrt(D)
    tbl_d, len_d = table.to(DEV), kv_lens.to(DEV)
    out = ops.paged_attention(q, kc, vc, None, len_d, tbl_d, 1, CTX, scale)
    assert torch.isfinite(out.float()).all()
    # (1) oracle on a sample of sequences (their pages gathered into a compact cache)
    sample = [0, 5, 77, 200, 255]
    pages = table.size(1)
    ids = table[sample].reshape(-1).long()
    kc_s, vc_s = kc[ids.to(DEV)].cpu(), vc[ids.to(DEV)].cpu()
    tbl_s = torch.arange(len(sample) * pages, dtype=torch.int32).view(len(sample), pages)
    ref = orc.paged_attention(q[sample].cpu(), kc_s, vc_s, torch.arange(len(sample) + 1, dtype=torch.int32),
                              kv_lens[sample], tbl_s, scale)
    assert rel_l2(out[sample], ref) <= 1e-3
    # (2) physical placement of the pages is irrelevant: move every page, same bits
    perm2 = torch.randperm(n_blocks, generator=g)
    inv = torch.empty_like(perm2)
    inv[perm2] = torch.arange(n_blocks)
    kc2, vc2 = kc[perm2.to(DEV)], vc[perm2.to(DEV)]                   # new block i holds old block perm2[i]
    out2 = ops.paged_attention(q, kc2, vc2, None, len_d, inv[table.long()].to(torch.int32).to(DEV), 1, CTX, scale)
    assert torch.equal(out2, out)
    # (3) sequences are independent: a batch of the sampled sequences alone gives the same rows (other launch plan)
    out3 = ops.paged_attention(q[sample], kc, vc, None, len_d[sample], tbl_d[sample].contiguous(), 1, CTX, scale)
    assert rel_l2(out3, out[sample]) <= 2e-4
    # (4) softmax weights form a convex combination: constant V rows come back (to rounding)
    vconst = torch.randn(NKV, D, generator=g).bfloat16().to(DEV)
    vc.copy_(vconst.expand_as(vc))
    out4 = ops.paged_attention(q, kc, vc, None, len_d, tbl_d, 1, CTX, scale).view(B, NQ, D)
    want = vconst.repeat_interleave(NQ // NKV, 0).expand(B, NQ, D).float()
    assert ((out4.float() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6).all()


def test_kv_write_full_size_round_trip():
    g = torch.Generator().manual_seed(101)
    T = 8192
    n_blocks, table = _paged_cache(g, 2, CTX)
    pos = torch.arange(CTX)
    slots = torch.cat([table[s].long()[pos // BS] * BS + pos % BS for s in range(2)]).to(torch.int32)
    k = torch.randn(T, NKV, D, generator=g).bfloat16().to(DEV)
    v = torch.randn(T, NKV, D, generator=g).bfloat16().to(DEV)
    kc = torch.zeros(n_blocks, BS, NKV, D, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.reshape_paged_cache(slots.to(DEV), k, v, kc, vc)
    rows = slots.long().to(DEV)
    assert torch.equal(kc.view(-1, NKV, D)[rows], k) and torch.equal(vc.view(-1, NKV, D)[rows], v)
    untouched = torch.ones(n_blocks * BS, dtype=torch.bool, device=DEV)
    untouched[rows] = False
    assert int(kc.view(-1, NKV * D)[untouched].abs().sum()) == 0


@pytest.mark.parametrize("M,N,K", [(256, 2 * I, H), (256, H, I), (8192, 2 * I, H), (8192, H, I), (8192, 4608, H)])
def test_int8_gemm_full_size_checksums(M, N, K):
    """exact int32 accumulators at the full Qwen2-7B shapes: column / row checksums over the WHOLE output (int64,
    exact), exact rows on a sample, and the fused dequant epilogue on the sample against the oracle formula"""
    gd = torch.Generator(device=DEV).manual_seed(M + N)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=DEV, generator=gd)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=gd)
    a_s = torch.rand(M, device=DEV, generator=gd) * 0.02 + 0.001
    w_s = torch.rand(N, device=DEV, generator=gd) * 0.02 + 0.001
    acc = torch.empty(M, N, dtype=torch.int32, device=DEV)
    out = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16, acc_out=acc)
    out_planned = ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)      # the kernel the planner really picks
    assert torch.equal(out_planned, out)
    a64, w64 = a.double(), w.double()                                      # |sums| < 2^53: exact in fp64
    assert torch.equal(acc.sum(0, dtype=torch.int64), (a64.sum(0) @ w64.T).to(torch.int64))   # per column
    assert torch.equal(acc.sum(1, dtype=torch.int64), (a64 @ w64.sum(0)).to(torch.int64))     # per row
    rows = torch.randint(0, M, (48,), device=DEV, generator=gd)
    exact = (a64[rows] @ w64.T).to(torch.int32)
    assert torch.equal(acc[rows], exact)
    ref = orc.scaled_matmul(a[rows].cpu(), w.cpu(), a_s[rows].cpu(), w_s.cpu(), torch.bfloat16, None) \
        if N * K <= 2 * I * H and M <= 256 else None
    want = (exact.float() * a_s[rows, None] * w_s[None, :]).bfloat16()    # same expression, same rounding
    assert torch.equal(out[rows], want)
    if ref is not None:
        assert torch.equal(out[rows].cpu(), ref)


@pytest.mark.parametrize("M,N,K,scales", [
    (256, 2 * I, H, "tc"), (256, H, I, "tc"), (8192, 2 * I, H, "tc"), (8192, H, I, "tensor"), (8192, 4608, H, "tc"),
    (128, 16 * 192, 1536, "tc"), (128, 7168, 16 * 128, "tensor"), (128, 7168, 18432 // 8, "tc"),   # DeepSeek-V3, one TP=8 rank
])
def test_fp8_gemm_full_size_against_fp64(M, N, K, scales):
    """fp8 e4m3 GEMM at the full Qwen2-7B and DeepSeek-V3 (one TP=8 rank) shapes. e4m3 x e4m3 products are exact in fp32, so
    the only freedom is the fp32 summation order: sampled rows against the fp64 product of the dequantised operands within a
    few accumulation ulps + one bf16 rounding (bar 2e-2 of BASELINE far away), and linear checksums of the WHOLE output
    (per column and per row) against fp64 sums computed without forming the reference product."""
    gd = torch.Generator(device=DEV).manual_seed(M * 7 + N)
    a = (torch.randn(M, K, device=DEV, generator=gd) * 2).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, device=DEV, generator=gd) * 0.5).to(torch.float8_e4m3fn)
    per = scales == "tc"                                   # per-token + per-channel, or one scale per tensor
    a_s = torch.rand(M if per else 1, device=DEV, generator=gd) * 0.05 + 0.01
    w_s = torch.rand(N if per else 1, device=DEV, generator=gd) * 0.02 + 0.01
    out = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.bfloat16)
    a64 = a.double() * (a_s.double()[:, None] if per else a_s.double())
    w64 = w.double() * (w_s.double()[:, None] if per else w_s.double())
    rows = torch.randint(0, M, (32,), device=DEV, generator=gd)
    ref = a64[rows] @ w64.T
    mag = (a64[rows].abs() @ w64.abs().T)                  # the error scale of a length-K fp32 sum
    err = (out[rows].double() - ref).abs()
    # measured (tools/fp8_fullsize_diag.py): the fp8 MFMA's running sum is a little coarser than a chain of fp32 FMAs -- worst
    # entry 2^-19.3 * sum|a||w| over 10^7 outputs (torch's fp32 product: 2^-27) -- still four orders below the 2e-2 bar
    assert (err <= 2.0 ** -8 * ref.abs() + 2.0 ** -17 * mag).all(), float((err / mag).max())
    assert rel_l2(out[rows], ref) <= 3e-3                  # bf16 output rounding only
    # whole-output checksums: sum_m out[m, n] = (sum_m a[m]) . w[n];  sum_n out[m, n] = a[m] . (sum_n w[n])
    col = a64.sum(0) @ w64.T
    row = a64 @ w64.sum(0)
    o64 = out.double()                                     # independent bf16 roundings add in quadrature (6 x 2^-9 rms > 6 sigma) + fp32 sums
    tol_c = 6 * 2.0 ** -9 * o64.pow(2).sum(0).sqrt() + 2.0 ** -19 * (a64.abs().sum(0) @ w64.abs().T) + 1e-9
    tol_r = 6 * 2.0 ** -9 * o64.pow(2).sum(1).sqrt() + 2.0 ** -19 * (a64.abs() @ w64.abs().sum(0)) + 1e-9
    assert ((o64.sum(0) - col).abs() <= tol_c).all()
    assert ((o64.sum(1) - row).abs() <= tol_r).all()


def test_silu_mul_quant_full_size():
    gd = torch.Generator(device=DEV).manual_seed(7)
    M = 8192
    x = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV).normal_(generator=gd) * 2
    q, s = ops.act_and_mul_dynamic_int8_quant(x, "silu")
    act = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    ops.act_and_mul(act, x, "silu")
    q2, s2 = ops.scaled_quantize(act)
    assert torch.equal(q, q2) and torch.equal(s, s2)                       # fusion == two operators, every row
    amax = act.float().abs().amax(-1)                                      # scale = amax / 127 on all rows
    torch.testing.assert_close(s, amax / 127.0, rtol=3e-7, atol=0)        # torch divides by reciprocal-multiply: 1 ulp
    assert int(q.abs().max()) == 127 and int((q.abs().amax(-1) != 127).sum()) == 0
    rows = [0, 1234, 8191]
    ref = torch.empty(len(rows), I, dtype=torch.bfloat16)
    orc.act_and_mul(ref, x[rows].cpu(), "silu")
    got = act[rows].cpu()
    assert (got == ref).float().mean() >= 0.995
    assert ((got.float() - ref.float()).abs() <= 2.0 ** -7 * ref.float().abs() + 1e-30).all()


def test_prefill_full_size_causality_and_sample():
    """two sequences of 4096 tokens (the bench's prefill chunk): (a) the last 48 queries of a sequence against the
    oracle (run as a chunked-prefill problem: 48 queries over 4096 keys, bottom-right causal), (b) causality: changing
    K/V of the last 1000 tokens cannot change the outputs of the earlier queries (bitwise)"""
    gd = torch.Generator(device=DEV).manual_seed(9)
    S, nseq = 4096, 2
    T = S * nseq
    qkv = torch.empty(T, (NQ + 2 * NKV) * D, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    q = qkv[:, :NQ * D].unflatten(-1, (NQ, D))
    k = qkv[:, NQ * D:(NQ + NKV) * D].unflatten(-1, (NKV, D))
    v = qkv[:, (NQ + NKV) * D:].unflatten(-1, (NKV, D))
    cu = torch.tensor([0, S, 2 * S], dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(D)
    out = ops.prefill_attention(q, k, v, cu, cu, S, scale, True).clone()
    tail = 48
    ref = orc.attention_varlen(q[S - tail:S].cpu().contiguous(), k[:S].cpu().contiguous(), v[:S].cpu().contiguous(),
                               torch.tensor([0, tail], dtype=torch.int32), torch.tensor([0, S], dtype=torch.int32),
                               scale, causal=True)
    # (one 16-bit P per score by default, as in the reference: bars relative to the effect of that rounding, see
    #  tests/test_gpu_parity.py::assert_prefill_close)
    ref1 = orc.attention_varlen(q[S - tail:S].cpu().contiguous(), k[:S].cpu().contiguous(), v[:S].cpu().contiguous(),
                                torch.tensor([0, tail], dtype=torch.int32), torch.tensor([0, S], dtype=torch.int32),
                                scale, causal=True, p_round=True)
    e_ref = rel_l2(ref1, ref)
    strict = os.environ.get("XLLM_MI355_PREFILL_P") == "2"
    assert rel_l2(out[S - tail:S], ref) <= (1e-3 if strict else max(1e-3, 1.25 * e_ref))
    if not strict and os.environ.get("XLLM_MI355_PREFILL_DMA", "1") != "0":
        # round 3: at BASELINE's context length the default kernel holds the ABSOLUTE 1e-3 against the oracle with its own cast
        # point (un-normalised 64-key-tile P rounded to bf16; oracle p_round="flash")
        ref2 = orc.attention_varlen(q[S - tail:S].cpu().contiguous(), k[:S].cpu().contiguous(), v[:S].cpu().contiguous(),
                                    torch.tensor([0, tail], dtype=torch.int32), torch.tensor([0, S], dtype=torch.int32),
                                    scale, causal=True, p_round="flash")
        assert rel_l2(out[S - tail:S], ref2) <= 1e-3, rel_l2(out[S - tail:S], ref2)
    cut = S - 1000
    qkv[cut:S, NQ * D:] = torch.empty(S - cut, 2 * NKV * D, dtype=torch.bfloat16, device=DEV).normal_(generator=gd)
    out2 = ops.prefill_attention(q, k, v, cu, cu, S, scale, True)
    assert torch.equal(out2[:cut], out[:cut])          # the past is untouched
    assert torch.equal(out2[S:], out[S:])              # and so is the other sequence
    assert not torch.equal(out2[cut:S], out[cut:S])


def test_mla_prefill_8192_tokens_against_the_flash_cast_point_oracle():
    """cfg4's context (configs[3]: DeepSeek-V3 MLA, ctx = 8192), one TP = 8 rank's 16 heads: the tile-sharing MLA prefill kernel
    (one 16-bit P per score, 64-token tiles, lazy running maximum) against the oracle with the same cast point (p_round="flash")
    at an ABSOLUTE 1e-3 on the last 24 queries of an 8192-token sequence (every key visible to them), plus the two relative bars
    that tie it to the reference's normalised-P spec (tests/_bars.py). Round-2 review, weak #1."""
    from _bars import assert_p16_attention_close
    g = torch.Generator().manual_seed(11)
    S, Hh, bs = 8192, 16, 64
    nb = S // bs + 2
    kc = torch.randn(nb, bs, 1, 576, generator=g).bfloat16()
    q = torch.randn(S, Hh, 576, generator=g).bfloat16()
    table = torch.randperm(nb, generator=g)[: S // bs].to(torch.int32).view(1, -1)
    cu_q = torch.tensor([0, S], dtype=torch.int32)
    kv_lens = torch.tensor([S], dtype=torch.int32)
    scale = 192 ** -0.5
    out = ops.mla_prefill(q.to(DEV), kc.to(DEV), cu_q.to(DEV), kv_lens.to(DEV), table.to(DEV), 512, scale, S, is_causal=True)
    tail = 24
    qt = q[S - tail:].contiguous()
    cu_t = torch.tensor([0, tail], dtype=torch.int32)
    refs = [orc.paged_attention(qt, kc, kc, cu_t, kv_lens, table, scale, causal=True, dv=512, p_round=m)
            for m in (False, True, "flash")]
    got = out[S - tail:].reshape(tail, -1)
    assert_p16_attention_close(got, refs[0].view(tail, -1), refs[1].view(tail, -1), refs[2].view(tail, -1))
