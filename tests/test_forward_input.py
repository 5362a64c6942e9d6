"""ForwardInput / the contiguous input buffer (runtime/forward_params.h:87-175, 421-470): layout arithmetic, packing, views."""
import pytest
import torch

from xllm_amd import attention
from xllm_amd.forward_input import (ForwardInputBufferPlan, align_up, forward_input_from_batch,
                                    kForwardInputBufferAlignment)


def _batch():
    lens, cached = [37, 64, 5, 21], [30, 0, 4, 21 - 3]
    bs = 16
    need = [(L + bs - 1) // bs for L in lens]
    blocks, nxt = [], 3
    for n in need:
        blocks.append(list(range(nxt, nxt + n)))
        nxt += n + 1
    bi = attention.build_batch_input(cached, lens, blocks, bs)
    T = int(bi.q_cu_seq_lens[-1])
    return bi, torch.arange(T, dtype=torch.int32) * 7 % 1000


def test_plan_layout_is_the_reference_arithmetic():
    assert align_up(0, 16) == 0 and align_up(1, 16) == 16 and align_up(16, 16) == 16 and align_up(17, 0) == 17
    plan = ForwardInputBufferPlan()
    got = {}
    tensors = [torch.arange(5, dtype=torch.int32), None, torch.arange(3, dtype=torch.int64), torch.zeros(0, dtype=torch.int32),
               torch.arange(7, dtype=torch.float32).view(7, 1), torch.tensor([True, False, True])]
    for i, t in enumerate(tensors):
        assert plan.add(t, (lambda k: lambda v: got.__setitem__(k, v))(i))
    assert len(plan.entries) == 5                                    # the undefined tensor takes no entry
    total = plan.prepare_layout()
    offs = [e["offset"] for e in plan.entries]
    assert offs == [0, 32, 64, 64, 96] and total == 112             # 20 -> 32, 24 -> 32, 0 -> 0, 28 -> 32, 3 -> 16
    assert all(o % kForwardInputBufferAlignment == 0 for o in offs)
    buf = plan.build_host_buffer(total, pin=False)
    assert buf.numel() == total and int(buf[20:32].sum()) == 0      # zero-filled tail of the first entry
    plan.bind_device_views(buf)
    for i, t in enumerate(tensors):
        if t is not None:
            assert got[i].dtype == t.dtype and got[i].shape == t.shape and torch.equal(got[i], t)
            if t.numel():
                assert got[i].untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()   # views of ONE buffer
    if torch.cuda.is_available():
        assert not ForwardInputBufferPlan().add(torch.zeros(2, device="cuda"), None)             # host tensors only


def test_forward_input_round_trip_on_the_host():
    bi, toks = _batch()
    fi = forward_input_from_batch(bi, toks, temperatures=torch.full((4,), 0.8))
    out = fi.to("cpu")
    a, b = fi.input_params.attention, out.input_params.attention
    for name in ("q_seq_lens", "kv_seq_lens", "q_cu_seq_lens", "new_cache_slots", "block_tables", "paged_kv_indptr",
                 "paged_kv_indices", "paged_kv_last_page_len"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(out.token_ids, toks) and torch.equal(out.positions, bi.positions)
    assert torch.equal(out.sampling_params.selected_token_idxes, (bi.q_cu_seq_lens[1:] - 1).to(torch.int32))
    assert torch.equal(out.sampling_params.temperatures, torch.full((4,), 0.8))
    assert out.device_tensors_ready and out.input_host_buffer.numel() % 16 == 0
    md = out.attention_metadata(is_prefill=False, is_chunked_prefill=True)
    ref = attention.build_attention_metadata(bi, False, True, "cpu")
    for name in ("q_cu_seq_lens", "kv_cu_seq_lens", "kv_seq_lens", "slot_mapping", "block_table"):
        assert torch.equal(getattr(md, name), getattr(ref, name)), name
    assert (md.max_query_len, md.max_seq_len, md.is_causal) == (ref.max_query_len, ref.max_seq_len, ref.is_causal)


@pytest.mark.gpu
def test_forward_input_reaches_the_device_in_one_copy_and_drives_a_model_step():
    from xllm_amd import layers
    from xllm_amd.attention import KVCache
    dev = "cuda"
    bi, toks = _batch()
    args = layers.ModelArgs(512, 2, 8, 2, 64, 1024, 1000, 1e-6, 1e4, 4096)
    model = layers.Qwen2Model(args, "int8", torch.bfloat16, dev, seed=3)
    nb = int(bi.block_tables.max()) + 2
    mk = lambda: [KVCache(torch.zeros(nb, 16, 2, 64, dtype=torch.bfloat16, device=dev),
                          torch.zeros(nb, 16, 2, 64, dtype=torch.bfloat16, device=dev)) for _ in model.layers]
    fi = forward_input_from_batch(bi, toks).to(dev)
    base = fi.input_device_buffer.untyped_storage().data_ptr()
    for t in (fi.token_ids, fi.input_params.attention.block_tables, fi.input_params.attention.new_cache_slots):
        assert t.is_cuda and t.untyped_storage().data_ptr() == base                 # views of the ONE device buffer
    assert fi.positions.dtype == torch.int64                                         # CUDA-branch normalisation
    md = fi.attention_metadata(is_prefill=False, is_chunked_prefill=True)
    kv_a, kv_b = mk(), mk()
    out = model.forward(fi.token_ids.long(), fi.positions, md, kv_a)
    ref_md = attention.build_attention_metadata(bi, False, True, dev)
    ref = model.forward(toks.to(dev).long(), bi.positions.to(dev).long(), ref_md, kv_b)
    assert torch.equal(out, ref) and torch.equal(kv_a[0].k_cache, kv_b[0].k_cache)


def test_native_plan_and_pack_entry_points():
    """xllm_mi355_host_plan_input_buffer / _pack_input_buffer (host code of the library, no GPU): the reference's arithmetic for
    several alignments, an undersized buffer is refused, empty entries take no bytes"""
    import ctypes as C
    from xllm_amd import _lib
    l = _lib.lib()
    payloads = [bytes(range(1, 21)), b"", bytes([7] * 3), bytes([9] * 64)]
    bufs = [C.create_string_buffer(p, max(len(p), 1)) for p in payloads]
    for alignment in (16, 1, 0, 64):
        arr = (_lib.HostBufferEntry * len(payloads))()
        for i, p in enumerate(payloads):
            arr[i].data, arr[i].bytes = C.cast(bufs[i], C.c_void_p).value if p else None, len(p)
        total = C.c_uint64(0)
        assert l.xllm_mi355_host_plan_input_buffer(arr, len(payloads), alignment, C.byref(total)) == 0
        off, want = 0, []
        for p in payloads:
            off = align_up(off, alignment)
            want.append((off, align_up(len(p), alignment)))
            off += want[-1][1]
        assert [(int(e.offset), int(e.aligned_bytes)) for e in arr] == want and total.value == off
        out = C.create_string_buffer(b"\xff" * (off + 8), off + 8)
        assert l.xllm_mi355_host_pack_input_buffer(arr, len(payloads), out, off) == 0
        raw = out.raw
        for p, (o, ab) in zip(payloads, want):
            assert raw[o:o + len(p)] == p and raw[o + len(p):o + ab] == b"\x00" * (ab - len(p))
        assert raw[off:] == b"\xff" * 8                                            # nothing written past the plan
        if off:
            assert l.xllm_mi355_host_pack_input_buffer(arr, len(payloads), out, off - 1) == -4   # XM_ERR_WORKSPACE
    assert l.xllm_mi355_host_plan_input_buffer(None, 0, 16, C.byref(total)) == 0 and total.value == 0
    assert l.xllm_mi355_host_plan_input_buffer(None, 3, 16, C.byref(total)) == -1   # XM_ERR_INVALID
