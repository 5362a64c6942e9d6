"""ctypes wrapper over oracle/liboracle.so (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by xllm_amd/ (the product path).

All tensors are torch CPU tensors (torch supplies bf16/fp16 storage); every wrapper
mirrors one xllm::kernel::* operator (xllm/core/kernels/ops_api.h) or one attention
mode (xllm/core/layers/dcu/torch_attention.cpp) -- see xllm_oracle.c for file:line.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "xllm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _dt(t: torch.Tensor) -> int:
    return _DT[t.dtype]


def _p(t):
    if t is None:
        return C.c_void_p(0)
    assert t.device.type == "cpu"
    return C.c_void_p(t.data_ptr())


def _i64(x):
    return C.c_int64(int(x))


def _f32(x):
    return C.c_float(float(x))


# --------------------------------------------------------------------------- fixtures
def seeded_tensor(key: str, shape, dtype=torch.bfloat16) -> torch.Tensor:
    """tests/core/layers/mlu/tests_utils.cpp:189-274 (FNV-1a -> SplitMix64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    if dtype in _DT:
        out = torch.empty(n, dtype=dtype)
        lib().orc_seeded_float(key.encode(), _i64(n), _p(out), C.c_int(_DT[dtype]))
    elif dtype == torch.int8:
        out = torch.empty(n, dtype=torch.int8)
        lib().orc_seeded_i8(key.encode(), _i64(n), _p(out))
    elif dtype == torch.int32:
        out = torch.empty(n, dtype=torch.int32)
        lib().orc_seeded_i32(key.encode(), _i64(n), _p(out))
    else:
        raise TypeError(dtype)
    return out.view(*shape)


def make_noise(key: str, shape, stddev: float, dtype=torch.bfloat16) -> torch.Tensor:
    """MakeNoise of tests/core/layers/mlu/qwen2_attention_test.cpp:112-121:
    (u - 0.5) * (sqrt(12) * stddev), evaluated with torch's 16-bit tensor arithmetic."""
    noise = seeded_tensor(key, shape, dtype)
    return (noise - 0.5) * (float(np.sqrt(np.float32(12.0))) * stddev)


def cache_slots(block_ids, block_size: int, pos_start: int, pos_end: int):
    """KVCacheState::cache_slots, framework/request/sequence_kv_state.cpp:86-104."""
    return [int(block_ids[i // block_size]) * block_size + i % block_size for i in range(pos_start, pos_end)]


def build_batch_metadata(seq_lens, q_lens, block_ids_per_seq, block_size: int):
    """BatchInputBuilder::setup_kv_cache_info + finalisation,
    framework/batch/batch_input_builder.cpp:739-830, 904-938:
    new_cache_slots, paged_kv_{indptr,indices,last_page_len}, 0-padded block_tables,
    cumulative q/kv seq lens with a leading 0."""
    slots, indptr, indices, last = [], [0], [], []
    for s, ql, blocks in zip(seq_lens, q_lens, block_ids_per_seq):
        n_cached = s - ql
        slots += cache_slots(blocks, block_size, n_cached, s)
        indices += [int(b) for b in blocks]
        indptr.append(indptr[-1] + len(blocks))
        last.append(block_size if s % block_size == 0 else s % block_size)
    width = max(len(b) for b in block_ids_per_seq) if block_ids_per_seq else 0
    table = np.zeros((len(seq_lens), width), dtype=np.int32)  # pad value 0 (batch_input_builder.cpp:934)
    for i, b in enumerate(block_ids_per_seq):
        table[i, : len(b)] = b
    i32 = lambda x: torch.tensor(np.asarray(x, dtype=np.int32))
    return dict(
        new_cache_slots=i32(slots),
        paged_kv_indptr=i32(indptr),
        paged_kv_indices=i32(indices),
        paged_kv_last_page_len=i32(last),
        block_tables=torch.from_numpy(table),
        kv_seq_lens=i32(seq_lens),
        q_seq_lens=i32(q_lens),
        q_cu_seq_lens=i32(np.concatenate([[0], np.cumsum(q_lens)])),
        kv_cu_seq_lens=i32(np.concatenate([[0], np.cumsum(seq_lens)])),
    )


def build_block_table_from_paged_kv(indptr, indices):
    B = indptr.numel() - 1
    total = indices.numel()
    table = torch.empty(B, total, dtype=torch.int32)
    lib().orc_build_block_table(_p(indptr), _p(indices), C.c_int32(B), C.c_int32(total), _p(table))
    return table


def decode_metadata_update(src: dict, dst: dict, n_tok, n_tok_padded, B, n_idx, B_padded=0):
    """CPU restatement of the N2 metadata refresh (same dict keys as xllm_amd.ops.decode_metadata_update)."""
    sk = ("tokens", "positions", "new_cache_slots", "kv_seq_lens", "paged_kv_indptr", "paged_kv_indices",
          "paged_kv_last_page_len")
    dk = ("tokens", "positions", "new_cache_slots", "kv_seq_lens", "kv_seq_lens_delta", "paged_kv_indptr",
          "paged_kv_indices", "paged_kv_last_page_len")
    bt = dst.get("block_table")
    args = [_p(src.get(k)) if src.get(k) is not None else None for k in sk]
    args += [_p(dst.get(k)) if dst.get(k) is not None else None for k in dk]
    args += [C.c_int64(n_tok), C.c_int64(n_tok_padded), C.c_int64(B), C.c_int64(n_idx)]
    args += [_p(bt) if bt is not None else None, _p(dst["kv_lens"]) if dst.get("kv_lens") is not None else None,
             C.c_int64(bt.size(1) if bt is not None else 0),
             C.c_int64(B_padded or (bt.size(0) if bt is not None else B))]
    lib().orc_decode_metadata_update(*args)


# --------------------------------------------------------------------------- operators
def reshape_paged_cache(slot_ids, k, v, k_cache, v_cache):
    """v / v_cache may both be None: K-only caches (MLA store_latent_cache, deepseek_v2_attention.cpp:170-178)"""
    T, nkv, d = k.shape[-3:]
    assert k.stride(-1) == 1 and k.stride(-2) == d
    assert (v is None) == (v_cache is None)
    if v is not None:
        assert v.stride(-1) == 1 and v.stride(-2) == d
    rc = lib().orc_reshape_paged_cache(
        _p(slot_ids), _p(k), _p(v), _p(k_cache), _p(v_cache), _i64(T), _i64(nkv), _i64(d),
        _i64(k_cache.shape[-3]), _i64(k_cache.shape[0]), _i64(k.stride(-3)),
        _i64(v.stride(-3) if v is not None else 0), C.c_int(k.element_size()))
    if rc:
        raise ValueError("slot out of range")


def block_copy(k_caches, v_caches, src_block_indices, dst_block_indices, cum_sum):
    """cuda::block_copy (kernels/cuda/block_copy.cu:38-118) on lists of per-layer caches [n_blocks, ...]: destination j copies
    from source group g = the first g with j < cum_sum[g] (the kernel's binary search, restated as written, :38-50);
    v_caches may be None (K-only caches). In place."""
    src, dst, cs = (t.tolist() for t in (src_block_indices, dst_block_indices, cum_sum))
    assert len(src) == len(cs)
    if not src:
        return
    for j, d in enumerate(dst):
        lo, hi = 0, len(src) - 1
        while lo < hi:
            mid = lo + ((hi - lo) >> 1)
            if j < cs[mid]:
                hi = mid
            else:
                lo = mid + 1
        for l in range(len(k_caches)):
            k_caches[l][d] = k_caches[l][src[lo]].clone()
            if v_caches is not None:
                v_caches[l][d] = v_caches[l][src[lo]].clone()


def rms_norm(out, x, w, eps):
    T, H = x.shape
    lib().orc_rms_norm(_p(out), _p(x), _p(w), _f32(eps), _i64(T), _i64(H), _i64(x.stride(0)), C.c_int(_dt(x)))


def fused_add_rms_norm(x, residual, w, eps):
    T, H = x.shape
    lib().orc_fused_add_rms_norm(_p(x), _p(residual), _p(w), _f32(eps), _i64(T), _i64(H),
                                 _i64(x.stride(0)), C.c_int(_dt(x)))


def rms_norm_static_fp8_quant(out_u8, x, w, scale, eps, residual=None):
    T, H = x.shape
    lib().orc_rms_norm_static_fp8_quant(_p(out_u8), _p(x), _p(residual), _p(w), _p(scale), _f32(eps),
                                        _i64(T), _i64(H), _i64(x.stride(0)), C.c_int(_dt(x)))


def build_cos_sin_cache(max_pos, rot_dim, theta, dtype):
    cache = torch.empty(max_pos, rot_dim, dtype=dtype)
    lib().orc_build_cos_sin_cache(_p(cache), _i64(max_pos), _i64(rot_dim), _f32(theta), C.c_int(_DT[dtype]))
    return cache


def rotary_embedding(positions, q, k, cos_sin_cache, head_size, is_neox=True):
    """q [T, nq*head_size] (or strided view), k optional; in place. positions int64."""
    T = q.shape[0]
    nq = q.shape[-1] // head_size if q.dim() == 2 else q.shape[-2]
    nk = 0 if k is None else (k.shape[-1] // head_size if k.dim() == 2 else k.shape[-2])
    positions = positions.to(torch.int64).contiguous()
    lib().orc_rotary_embedding(
        _p(positions), _p(q), _p(k), _p(cos_sin_cache), _i64(T), _i64(nq), _i64(nk), _i64(head_size),
        _i64(cos_sin_cache.shape[-1]), _i64(q.stride(0)), _i64(0 if k is None else k.stride(0)),
        _i64(head_size), C.c_int(1 if is_neox else 0), C.c_int(_dt(q)))


_ACT = {"silu": 0, "gelu": 1, "gelu_tanh": 2}


def act_and_mul(out, x, mode="silu"):
    T, d2 = x.shape
    lib().orc_act_and_mul(_p(out), _p(x), _i64(T), _i64(d2 // 2), C.c_int(_ACT[mode]), C.c_int(_dt(x)))


def scaled_quantize(x):
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.int8)
    s = torch.empty(M, dtype=torch.float32)
    x_c = x.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_scaled_quantize_i8(_p(x_c), _p(q), _p(s), _i64(M), _i64(K), C.c_int(_dt(x)))
    return q, s


def scaled_matmul(a, w, a_scale, w_scale, out_dtype=torch.bfloat16, bias=None, want_acc=False):
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=out_dtype)
    acc = torch.empty(M, N, dtype=torch.int32) if want_acc else None
    a_scale_c = a_scale.contiguous()  # keep the (possibly new) tensor alive across the call
    w_scale_c = w_scale.reshape(-1).contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_scaled_matmul_i8(_p(a), _p(w), _p(a_scale_c), _p(w_scale_c),
                               _p(bias), _p(out), _p(acc), _i64(M), _i64(N), _i64(K), C.c_int(_DT[out_dtype]))
    return (out, acc) if want_acc else out


def matmul(a, w, bias=None):
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=a.dtype)
    a_c = a.contiguous()  # keep the (possibly new) tensor alive across the call
    w_c = w.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_matmul(_p(a_c), _p(w_c), _p(bias), _p(out), _i64(M), _i64(N), _i64(K),
                     C.c_int(_dt(a)))
    return out


def bmm_heads(x, w_nk):
    """torch::bmm(x.transpose(0, 1), w).transpose(0, 1) of DeepseekV2AttentionImpl (layers/dcu/deepseek_v2_attention.cpp:180-187,
    310-311) with w given as [h, N, K] (K contiguous): per head F::linear, fp32 sequential sums, one rounding (oracle matmul)."""
    T, H, K = x.shape
    return torch.stack([matmul(x[:, h].contiguous(), w_nk[h].contiguous(), None) for h in range(H)], 1)


def static_scaled_fp8_quant(x, scale):
    out = torch.empty(x.shape, dtype=torch.uint8)
    x_c = x.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_static_scaled_fp8_quant(_p(out), _p(x_c), _p(scale), _i64(x.numel()), C.c_int(_dt(x)))
    return out


def fp8_scaled_quantize(x, scale=None):
    if scale is None:
        scale = torch.empty(1, dtype=torch.float32)
        x_c = x.contiguous()  # keep the (possibly new) tensor alive across the call
        lib().orc_fp8_dynamic_scale(_p(x_c), _i64(x.numel()), C.c_int(_dt(x)), _p(scale))
    return static_scaled_fp8_quant(x, scale), scale


def fp8_scaled_matmul(a_u8, w_u8, a_scale, w_scale, out_dtype=torch.bfloat16, bias=None):
    M, K = a_u8.shape
    N = w_u8.shape[0]
    out = torch.empty(M, N, dtype=out_dtype)
    lib().orc_fp8_scaled_matmul(_p(a_u8), _p(w_u8), _p(a_scale), _i64(a_scale.numel()), _p(w_scale),
                                _i64(w_scale.numel()), _p(bias), _p(out), _i64(M), _i64(N), _i64(K),
                                C.c_int(_DT[out_dtype]))
    return out


def e4m3_to_f32(u8):
    out = torch.empty(u8.shape, dtype=torch.float32)
    u8_c = u8.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_e4m3_to_f32(_p(u8_c), _p(out), _i64(u8.numel()))
    return out


def _p_mode(p_round) -> int:
    """False / 0: P stays fp32; True / 1: the NORMALISED P is rounded to the tensor dtype before PV (the reference's eager spec,
    flashinfer_attention.cpp:84-90); "flash" / 2: the un-normalised tile P of a flash kernel is rounded (64-key tiles, lazy
    running maximum: xllm_oracle.c::attn_one_query_flash)"""
    if p_round == "flash":
        return 2
    return int(p_round)


def attention_varlen(q, k, v, cu_q, cu_k, scale, causal=True, window_left=-1, p_round=False):
    """q [Tq, nq, d], k/v [Tk, nkv, d] (token-strided views allowed) -> out [Tq, nq*d]."""
    Tq, nq, d = q.shape
    nkv = k.shape[1]
    out = torch.empty(Tq, nq * d, dtype=q.dtype)
    rc = lib().orc_attention_varlen(
        _p(q), _p(k), _p(v), _p(out), _p(cu_q), _p(cu_k), _i64(cu_q.numel() - 1), _i64(nq), _i64(nkv),
        _i64(d), _i64(q.stride(0)), _i64(k.stride(0)), _i64(v.stride(0)), _f32(scale),
        C.c_int(int(causal)), _i64(window_left), C.c_int(_dt(q)), C.c_int(_p_mode(p_round)))
    if rc:
        raise ValueError(f"orc_attention_varlen rc={rc}")
    return out


def paged_attention(q, k_cache, v_cache, cu_q, kv_lens, block_table, scale, causal=False,
                    window_left=-1, p_round=False, dv=None):
    """q [Tq, nq, d]; caches [n_blocks, bs, nkv, d]; -> out [Tq, nq*dv]."""
    Tq, nq, d = q.shape
    n_blocks, bs, nkv, dc = k_cache.shape
    assert dc == d
    dv = d if dv is None else dv
    out = torch.empty(Tq, nq * dv, dtype=q.dtype)
    block_table = block_table.contiguous()
    rc = lib().orc_paged_attention(
        _p(q), _p(k_cache), _p(v_cache), _p(out), _p(cu_q), _p(kv_lens), _p(block_table),
        _i64(block_table.shape[1]), _i64(kv_lens.numel()), _i64(nq), _i64(nkv), _i64(d), _i64(dv),
        _i64(bs), _i64(n_blocks), _i64(q.stride(0)), _f32(scale), C.c_int(int(causal)), _i64(window_left),
        C.c_int(_dt(q)), C.c_int(_p_mode(p_round)))
    if rc:
        raise ValueError(f"orc_paged_attention rc={rc}")
    return out


def fused_qk_norm_rope(qkv, nq, nk, nv, d, eps, qw, kw, cos_sin, interleaved, positions):
    positions_c = positions.to(torch.int64).contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_fused_qk_norm_rope(_p(qkv), _i64(qkv.shape[0]), _i64(nq), _i64(nk), _i64(nv), _i64(d), _f32(eps),
                                 _p(qw), _p(kw), _p(cos_sin), C.c_int(_dt(cos_sin)), C.c_int(int(interleaved)),
                                 _p(positions_c), C.c_int(_dt(qkv)))


def moe_compute_index(expert_id, E):
    T, topk = expert_id.shape
    src_dst = torch.empty(T * topk, dtype=torch.int32)
    dst_src = torch.empty(T * topk, dtype=torch.int32)
    sizes = torch.empty(E, dtype=torch.int32)
    expert_id_c = expert_id.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_moe_compute_index(_p(expert_id_c), _i64(T), _i64(topk), _i64(E), _p(src_dst),
                                _p(dst_src), _p(sizes))
    return src_dst, dst_src, sizes


def moe_combine(gemm2, w, T, topk):
    H = gemm2.shape[-1]
    out = torch.empty(T, H, dtype=gemm2.dtype)
    gemm2_c = gemm2.contiguous()  # keep the (possibly new) tensor alive across the call
    w_c = w.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_moe_combine(_p(out), _p(gemm2_c), _p(w_c), _i64(T), _i64(topk), _i64(H),
                          C.c_int(_dt(gemm2)))
    return out


def group_gemm(a, w, token_count):
    E, N, K = w.shape
    out = torch.zeros(a.shape[0], N, dtype=a.dtype)
    a_c = a.contiguous()  # keep the (possibly new) tensor alive across the call
    w_c = w.contiguous()  # keep the (possibly new) tensor alive across the call
    lib().orc_group_gemm(_p(a_c), _p(w_c), _p(token_count), _p(out), _i64(E), _i64(N),
                         _i64(K), C.c_int(_dt(a)))
    return out


# --------------------------------------------------------------------------- N3 sampler
def philox_uniform(n, seed, offset):
    out = torch.empty(n, dtype=torch.float32)
    lib().orc_philox_uniform(_p(out), _i64(n), C.c_uint64(seed), C.c_uint64(offset))
    return out


def random_sample(probs, u):
    """probs [B, V] fp32, u [B] fp32 -> int32 [B] (fp64 CDF)"""
    B, V = probs.shape
    out = torch.empty(B, dtype=torch.int32)
    pc, uc = probs.contiguous(), u.contiguous()
    lib().orc_random_sample(_p(pc), _p(uc), _i64(B), _i64(V), _p(out))
    return out


def rejection_sample(draft_token_ids, num_draft_tokens, cu_num_draft_tokens, draft_probs, target_probs,
                     bonus_token_ids, uniform_rand, uniform_probs):
    B, V = num_draft_tokens.numel(), target_probs.size(1)
    out = torch.empty(B + draft_token_ids.numel(), dtype=torch.int32)
    dp, tp, ur, up = (t.contiguous() for t in (draft_probs, target_probs, uniform_rand, uniform_probs))
    lib().orc_rejection_sample(_p(draft_token_ids), _p(num_draft_tokens), _p(cu_num_draft_tokens),
                               _p(dp), _p(tp), _p(bonus_token_ids), _p(ur), _p(up), _i64(B), _i64(V), _p(out))
    return out


def moe_fused_topk(gating, topk, renormalize, correction_bias=None, scoring_func="softmax"):
    T, E = gating.shape
    w = torch.empty(T, topk, dtype=torch.float32)
    ids = torch.empty(T, topk, dtype=torch.int32)
    b = correction_bias.float().contiguous() if (correction_bias is not None and scoring_func == "sigmoid") else None
    g32 = gating.float().contiguous()   # keep the converted copy alive across the call (ctypes only sees its address)
    lib().orc_moe_fused_topk(_p(g32), _i64(T), _i64(E), _i64(topk), C.c_int(int(renormalize)),
                             _p(b), C.c_int(int(scoring_func == "sigmoid")), _p(w), _p(ids))
    return w, ids


def moe_grouped_topk(gating, topk, num_expert_group, topk_group, renormalize, correction_bias=None, scoring_func="softmax",
                     routed_scaling_factor=1.0):
    """dcu::moe_grouped_topk (kernels/dcu/topk_gate.cpp:59-125 -> aiter grouped_topk / biased_grouped_topk, an external
    library absent from the reference tree): the published DeepSeek-V2 / V3 grouped gate, pinned on the statistics of
    tests/core/layers/mlu/moe_gate_test.cpp:143-272 (tests/test_reference_fixtures.py)"""
    T, E = gating.shape
    assert E % num_expert_group == 0 and topk <= topk_group * (E // num_expert_group)
    assert correction_bias is None or scoring_func == "sigmoid"
    w = torch.empty(T, topk, dtype=torch.float32)
    ids = torch.empty(T, topk, dtype=torch.int32)
    b = correction_bias.float().contiguous() if correction_bias is not None else None
    g32 = gating.float().contiguous()
    lib().orc_moe_grouped_topk(_p(g32), _i64(T), _i64(E), _i64(topk), _i64(num_expert_group), _i64(topk_group),
                               C.c_int(int(renormalize)), _p(b), C.c_int(int(scoring_func == "sigmoid")),
                               C.c_float(float(routed_scaling_factor)), _p(w), _p(ids))
    return w, ids
