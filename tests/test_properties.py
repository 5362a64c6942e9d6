"""Property-based checks (hypothesis): random geometries that the hand-picked parametrisations do not enumerate.
CPU: the host batch builder against the oracle + structural invariants. GPU: paged decode / chunked prefill attention and
the MoE index build against the oracle on random ragged shapes. Bars as in test_gpu_parity.py."""
import math

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import oracle as orc
from xllm_amd import attention

COMMON = dict(deadline=None, suppress_health_check=list(HealthCheck))


@st.composite
def batches(draw, max_b=10, max_len=300):
    bs = draw(st.sampled_from([1, 4, 16, 64, 128]))
    B = draw(st.integers(1, max_b))
    lens = [draw(st.integers(1, max_len)) for _ in range(B)]
    q = [draw(st.integers(0, L)) for L in lens]
    spare = [draw(st.integers(0, 2)) for _ in range(B)]
    seed = draw(st.integers(0, 2 ** 16))
    return bs, lens, q, spare, seed


def _blocks(bs, lens, spare, seed):
    need = [(L + bs - 1) // bs + s for L, s in zip(lens, spare)]
    perm = np.random.default_rng(seed).permutation(sum(need) + 5).tolist()
    out, used = [], 0
    for n in need:
        out.append(perm[used:used + n]); used += n
    return out, sum(need) + 5


@settings(max_examples=60, **COMMON)
@given(batches())
def test_host_batch_builder_matches_oracle_on_random_batches(case):
    bs, lens, q, spare, seed = case
    blocks, _ = _blocks(bs, lens, spare, seed)
    cached = [L - x for L, x in zip(lens, q)]
    got = attention.build_batch_input(cached, lens, blocks, bs)
    ref = orc.build_batch_metadata(lens, q, blocks, bs)
    for k in ("new_cache_slots", "paged_kv_indptr", "paged_kv_indices", "paged_kv_last_page_len", "block_tables",
              "q_cu_seq_lens", "kv_cu_seq_lens", "kv_seq_lens", "q_seq_lens"):
        assert torch.equal(getattr(got, k), ref[k]), k
    slots = got.new_cache_slots.tolist()
    assert len(set(slots)) == len(slots)                      # no two new tokens share a cache slot
    t = 0
    for b, (c, L) in enumerate(zip(cached, lens)):
        for p in range(c, L):
            assert slots[t] // bs == blocks[b][p // bs] and slots[t] % bs == p % bs and int(got.positions[t]) == p
            t += 1
    assert int(got.q_cu_seq_lens[-1]) == sum(q) and int(got.kv_cu_seq_lens[-1]) == sum(lens)


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


@pytest.mark.gpu
@settings(max_examples=12, **COMMON)
@given(batches(max_b=6, max_len=700), st.sampled_from([(28, 4, 128), (14, 2, 64), (8, 8, 128), (7, 1, 128)]))
def test_paged_attention_random_geometry(case, heads):
    """decode (q = 1) and chunked prefill (ragged q chunks, bottom-right causal) over randomly placed pages"""
    from xllm_amd import ops
    bs, lens, q, spare, seed = case
    if bs < 4:
        bs = 16
    nq, nkv, d = heads
    blocks, nb = _blocks(bs, lens, spare, seed)
    g = torch.Generator().manual_seed(seed)
    kc = torch.randn(nb, bs, nkv, d, generator=g).bfloat16()
    vc = torch.randn(nb, bs, nkv, d, generator=g).bfloat16()
    scale = 1.0 / math.sqrt(d)
    for mode_q in ([1] * len(lens), [max(1, x) for x in q]):
        md = orc.build_batch_metadata(lens, mode_q, blocks, bs)
        T = sum(mode_q)
        qt = torch.randn(T, nq, d, generator=g).bfloat16()
        causal = max(mode_q) > 1
        ref = orc.paged_attention(qt, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale, causal=causal)
        out = ops.paged_attention(qt.cuda(), kc.cuda(), vc.cuda(), md["q_cu_seq_lens"].cuda() if causal else None,
                                  md["kv_seq_lens"].cuda(), md["block_tables"].cuda(), max(mode_q), max(lens), scale, causal)
        if causal:   # chunked prefill keeps one 16-bit P like the reference: bars relative to the p_round oracle
            from _bars import assert_p16_attention_close
            ref16 = orc.paged_attention(qt, kc, vc, md["q_cu_seq_lens"], md["kv_seq_lens"], md["block_tables"], scale,
                                        causal=True, p_round=True)
            assert_p16_attention_close(out.view(T, -1), ref.view(T, -1), ref16.view(T, -1))
        else:
            assert _rel(out.view(T, -1), ref.view(T, -1)) <= 1e-3


@pytest.mark.gpu
@settings(max_examples=15, **COMMON)
@given(st.integers(1, 3000), st.sampled_from([1, 2, 4, 8]), st.sampled_from([3, 8, 64, 128, 257]), st.integers(0, 2 ** 16))
def test_moe_index_build_random(T, topk, E, seed):
    from xllm_amd import ops
    g = torch.Generator().manual_seed(seed)
    topk = min(topk, E)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(min(T, 400))]).to(torch.int32)
    ids = ids.repeat((T + ids.size(0) - 1) // ids.size(0), 1)[:T].contiguous()
    got = ops.moe_compute_index(ids.cuda(), E)
    ref = orc.moe_compute_index(ids, E)
    for a, b in zip(got, ref):
        assert torch.equal(a.cpu(), b)


@pytest.mark.gpu
@settings(max_examples=8, **COMMON)
@given(st.integers(1, 900), st.sampled_from([1, 2, 4]), st.sampled_from([2, 7, 32]), st.sampled_from([128, 384]),
       st.sampled_from([8, 72, 256]), st.integers(0, 2 ** 16))
def test_group_gemm_w8a8_random(T, topk, E, K, N, seed):
    """W8A8 grouped GEMM (tile table + gather) == the oracle's scaled_matmul per expert, bit for bit, on random routing
    (empty experts, experts smaller than a tile, N smaller than a tile)"""
    from xllm_amd import ops
    g = torch.Generator().manual_seed(seed)
    topk = min(topk, E)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(T)]).to(torch.int32)
    if E > 2:
        ids[ids == 1] = 0                                   # expert 1 stays empty
    src_dst, dst_src, sizes = ops.moe_compute_index(ids.cuda(), E)
    xq = torch.randint(-127, 128, (T, K), generator=g, dtype=torch.int8)
    xs = torch.rand(T, generator=g) * 0.05 + 0.001
    wq = torch.randint(-127, 128, (E, N, K), generator=g, dtype=torch.int8)
    ws = torch.rand(E, N, generator=g) * 0.02 + 0.001
    tok = dst_src.cpu().long() // topk
    ref = torch.empty(T * topk, N, dtype=torch.bfloat16)
    off = 0
    for e in range(E):
        c = int(sizes[e])
        if c:
            ref[off:off + c] = orc.scaled_matmul(xq[tok[off:off + c]].contiguous(), wq[e], xs[tok[off:off + c]].contiguous(),
                                                 ws[e], torch.bfloat16, None)
        off += c
    got = ops.group_gemm_w8a8(xq.cuda(), xs.cuda(), wq.cuda(), ws.cuda(), sizes, row_index=dst_src, index_div=topk)
    assert torch.equal(got.cpu(), ref)
