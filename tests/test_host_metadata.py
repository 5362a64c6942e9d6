"""SURVEY 8 row a1 on the HOST: the product's BatchInputBuilder / AttentionMetadata mirror (C++ behind the C ABI, no GPU
involved) against the oracle's restatement of framework/batch/batch_input_builder.cpp and sequence_kv_state.cpp.
Integer work: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from xllm_amd import _lib, attention


def _case(seed, B, bs, max_len, decode=False):
    rng = np.random.default_rng(seed)
    seq_lens, cached, blocks, used = [], [], [], 0
    pool = rng.permutation(16384).tolist()
    for _ in range(B):
        L = int(rng.integers(1, max_len + 1))
        q = 1 if decode else int(rng.integers(1, L + 1))
        n = (L + bs - 1) // bs + int(rng.integers(0, 3))          # spare pages are legal
        seq_lens.append(L); cached.append(L - q)
        blocks.append(pool[used:used + n]); used += n
    return cached, seq_lens, blocks


@pytest.mark.parametrize("seed,B,bs,max_len,decode", [(0, 1, 16, 40, False), (1, 7, 16, 100, False), (2, 64, 128, 700, True),
                                                      (3, 33, 64, 64, False), (4, 5, 1, 9, False), (5, 256, 128, 4096, True)])
def test_batch_input_equals_oracle(seed, B, bs, max_len, decode):
    cached, seq_lens, blocks = _case(seed, B, bs, max_len, decode)
    got = attention.build_batch_input(cached, seq_lens, blocks, bs)
    q_lens = [L - c for L, c in zip(seq_lens, cached)]
    ref = orc.build_batch_metadata(seq_lens, q_lens, blocks, bs)
    for k_got, k_ref in [("new_cache_slots",) * 2, ("paged_kv_indptr",) * 2, ("paged_kv_indices",) * 2,
                         ("paged_kv_last_page_len",) * 2, ("block_tables",) * 2, ("q_cu_seq_lens",) * 2,
                         ("kv_cu_seq_lens",) * 2, ("kv_seq_lens",) * 2, ("q_seq_lens",) * 2]:
        assert torch.equal(getattr(got, k_got), ref[k_ref]), k_got
    assert torch.equal(got.positions, torch.cat([torch.arange(c, L, dtype=torch.int32) for c, L in zip(cached, seq_lens)]))
    assert got.q_max_seq_len == max(q_lens) and got.kv_max_seq_len == max(seq_lens) and got.total_kv_len == sum(seq_lens)
    # slot = block_id * block_size + pos % block_size, per sequence, through the single-sequence entry point too
    t = 0
    for c, L, b in zip(cached, seq_lens, blocks):
        one = attention.cache_slots(b, bs, c, L)
        assert torch.equal(one, torch.tensor(orc.cache_slots(b, bs, c, L), dtype=torch.int32))
        assert torch.equal(one, got.new_cache_slots[t:t + L - c]); t += L - c


def test_attention_metadata_flags_and_dummy_batch():
    cached, seq_lens, blocks = _case(9, 6, 16, 90)
    bi = attention.build_batch_input(cached, seq_lens, blocks, 16)
    md = attention.build_attention_metadata(bi, is_prefill=False, is_chunked_prefill=True, device="cpu")
    assert md.is_causal and not md.is_prefill and md.max_query_len == bi.q_max_seq_len
    assert torch.equal(md.kv_seq_lens, torch.diff(md.kv_cu_seq_lens)) and md.kv_cu_seq_lens[0] == 0
    assert torch.equal(md.block_table, bi.block_tables) and int((md.block_table[0, len(blocks[0]):] != 0).sum()) == 0
    md = attention.build_attention_metadata(bi, is_prefill=False, is_chunked_prefill=False, device="cpu")
    assert not md.is_causal                                         # decode: attention_metadata_builder.cpp:236-241
    # a step in which no sequence has a query token is the reference's dummy batch (:214-232)
    dummy = attention.build_batch_input([5, 17], [5, 17], [[3], [1, 2]], 16)
    assert dummy.q_max_seq_len == 0 and dummy.new_cache_slots.numel() == 0
    md = attention.build_attention_metadata(dummy, False, False, "cpu")
    assert md.q_cu_seq_lens.tolist() == [0, 1] and md.slot_mapping.tolist() == [1] and md.max_query_len == 1


def test_batch_input_rejects_what_the_reference_check_fails_on():
    with pytest.raises(_lib.Mi355Error):
        attention.build_batch_input([0], [40], [[7, 8]], 16)        # 40 tokens do not fit 2 pages of 16
    with pytest.raises(_lib.Mi355Error):
        attention.build_batch_input([0], [4], [[]], 16)             # "no cache blocks available"
    with pytest.raises(_lib.Mi355Error):
        attention.build_batch_input([9], [4], [[1]], 16)            # more cached tokens than tokens
    with pytest.raises(_lib.Mi355Error):
        attention.cache_slots([1, 2], 16, 0, 33)
    empty = attention.build_batch_input([], [], [], 16)
    assert empty.q_cu_seq_lens.tolist() == [0] and empty.block_tables.shape == (0, 0)


def test_batch_builder_reuses_buffers_and_matches_the_one_shot_builder():
    cached, seq_lens, blocks = _case(21, 9, 16, 200, decode=True)
    bb = attention.BatchBuilder(blocks, 16, max_tokens=9)
    for step in range(3):
        c = [x + step for x in cached]
        L = [x + step for x in seq_lens]
        if any(n > len(b) * 16 for n, b in zip(L, blocks)):
            break
        got, ref = bb.build(c, L), attention.build_batch_input(c, L, blocks, 16)
        for k in ("new_cache_slots", "positions", "paged_kv_indptr", "paged_kv_indices", "paged_kv_last_page_len",
                  "block_tables", "q_cu_seq_lens", "kv_cu_seq_lens", "q_seq_lens", "kv_seq_lens"):
            assert torch.equal(getattr(got, k), getattr(ref, k)), k
        assert (got.q_max_seq_len, got.kv_max_seq_len, got.total_kv_len) == (ref.q_max_seq_len, ref.kv_max_seq_len, ref.total_kv_len)
