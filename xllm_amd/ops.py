"""Host-side mirror of the reference's operator boundary for the hot path.

Names, argument meaning and error behaviour follow xllm::kernel::* (xllm/core/kernels/ops_api.h:27-287,
param.h) and the per-backend headers kernels/cuda/cuda_ops_api.h / kernels/dcu/dcu_ops_api.h; each function
validates like the reference's CHECKs, allocates outputs with torch.empty on the input's device when the
reference does, and calls ONE C-ABI symbol of include/xllm_mi355.h on torch's current HIP stream.
PyTorch is only the allocator / stream provider here.
"""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes as C
import os

import torch

from . import _lib
from ._lib import Mi355Error, check

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_ACT = {"silu": 0, "gelu": 1, "gelu_tanh": 2}
FP8 = torch.float8_e4m3fn


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    if t.dtype not in _DT:
        raise Mi355Error(f"unsupported dtype {t.dtype}")
    return _DT[t.dtype]


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise Mi355Error("xllm_amd ops need device tensors (no CPU fallback on the product path)")


# ------------------------------------------------------------------------------------------------ KV write
def reshape_paged_cache(slot_ids, key, value, key_cache, value_cache) -> None:
    """kernel::reshape_paged_cache(ReshapePagedCacheParams&) (ops_api.h:31, reshape_paged_cache.cu:65-100)."""
    if (value is None) != (value_cache is None):
        raise Mi355Error("value and value_cache must both be given or both be None (K-only cache)")
    _need_cuda(*[t for t in (slot_ids, key, value, key_cache, value_cache) if t is not None])
    if not (key.stride(-1) == 1 and key.stride(-2) == key.size(-1)) or (
            value is not None and not (value.stride(-1) == 1 and value.stride(-2) == value.size(-1))):
        raise Mi355Error("keys/values must be contiguous over (n_kv_heads, head_dim)")  # reference CHECK :73-74
    if slot_ids.dtype != torch.int32:
        raise Mi355Error("slot_ids must be int32")
    n_tokens, n_kv, d = key.shape[-3:]
    check(_lib.lib().xllm_mi355_reshape_paged_cache(
        _p(slot_ids), _p(key), _p(value), _p(key_cache), _p(value_cache), n_tokens, n_kv, d,
        key_cache.size(-3), key_cache.size(0), key.stride(-3), value.stride(-3) if value is not None else 0,
        key.element_size(), _stream()), "reshape_paged_cache")


def block_copy(key_cache_ptrs, value_cache_ptrs, src_block_indices, dst_block_indices, cum_sum, numel_per_block,
               cache_dtype) -> None:
    """xllm::kernel::cuda::block_copy (kernels/cuda/cuda_ops_api.h:50-56, block_copy.cu:120-205): key_cache_ptrs /
    value_cache_ptrs are int64 DEVICE tensors [num_layers] of cache base addresses (WorkerImpl::refresh_cuda_block_copy_runtime_state,
    runtime/worker_impl.cpp:1007-1058); destination j copies from source group g = first g with j < cum_sum[g].
    value_cache_ptrs may be None (K-only caches)."""
    if src_block_indices.numel() == 0:
        return                                                                       # block_copy.cu:128-130
    _need_cuda(*[t for t in (key_cache_ptrs, value_cache_ptrs, src_block_indices, dst_block_indices, cum_sum) if t is not None])
    if key_cache_ptrs.dtype != torch.int64 or (value_cache_ptrs is not None and value_cache_ptrs.dtype != torch.int64):
        raise Mi355Error("cache pointer tensors must be int64")                      # reference CHECK_EQ :141-142
    for t in (src_block_indices, dst_block_indices, cum_sum):
        if t.dtype != torch.int32 or t.dim() != 1 or not t.is_contiguous():
            raise Mi355Error("src_block_indices / dst_block_indices / cum_sum must be contiguous 1-D int32")
    if value_cache_ptrs is not None and value_cache_ptrs.numel() != key_cache_ptrs.numel():
        raise Mi355Error("key_cache_ptrs and value_cache_ptrs differ in length")     # :156
    if src_block_indices.numel() != cum_sum.numel() or numel_per_block <= 0:
        raise Mi355Error("src_block_indices and cum_sum must have one entry per source group; numel_per_block > 0")
    esz = torch.empty((), dtype=cache_dtype).element_size()
    check(_lib.lib().xllm_mi355_block_copy(_p(key_cache_ptrs), _p(value_cache_ptrs), _p(src_block_indices),
                                           _p(dst_block_indices), _p(cum_sum), key_cache_ptrs.numel(),
                                           src_block_indices.numel(), dst_block_indices.numel(),
                                           int(numel_per_block) * esz, _stream()), "block_copy")


def store_latent_cache(latent_cache, slot_mapping, k_cache) -> None:
    """DeepseekV2AttentionImpl::store_latent_cache (layers/dcu/deepseek_v2_attention.cpp:170-178):
    k_cache.view(-1, 576).index_copy_(0, slot_mapping, latent_cache) -- the K-only form of reshape_paged_cache."""
    reshape_paged_cache(slot_mapping, latent_cache.view(latent_cache.size(0), 1, latent_cache.size(-1)), None, k_cache,
                        None)


def build_block_table_from_paged_kv(paged_kv_indptr, paged_kv_indices) -> torch.Tensor:
    """dcu::build_block_table_from_paged_kv_cuda (dcu_ops_api.h, build_block_table_from_paged_kv.hip:74-110)."""
    _need_cuda(paged_kv_indptr, paged_kv_indices)
    B = paged_kv_indptr.numel() - 1
    total = paged_kv_indices.numel()
    table = torch.empty(B, total, dtype=torch.int32, device=paged_kv_indptr.device)
    check(_lib.lib().xllm_mi355_build_block_table_from_paged_kv(_p(paged_kv_indptr), _p(paged_kv_indices), B, total,
                                                                _p(table), _stream()), "build_block_table")
    return table


# ------------------------------------------------------------------------------------------------ norms
def rms_norm(output, input, weight, eps: float) -> None:
    """cuda::rms_norm(out, in, w, eps) (cuda_ops_api.h:146-150, norm.cu:430-464)."""
    _need_cuda(output, input, weight)
    H = input.size(-1)
    x2 = input.view(-1, H) if input.is_contiguous() else input
    if x2.dim() != 2 or x2.stride(-1) != 1 or not output.is_contiguous():
        raise Mi355Error("rms_norm: input must be [T, H] with unit inner stride, output contiguous")
    check(_lib.lib().xllm_mi355_rms_norm(_p(output), _p(x2), _p(weight), eps, x2.size(0), H, x2.stride(0), _dt(input),
                                        _stream()), "rms_norm")


def fused_add_rms_norm(input, residual, weight, eps: float) -> None:
    """cuda::fused_add_rms_norm(in, residual, w, eps) (norm.cu:466-512): both updated in place."""
    _need_cuda(input, residual, weight)
    H = input.size(-1)
    if not (input.is_contiguous() and residual.is_contiguous()):
        raise Mi355Error("fused_add_rms_norm: contiguous input/residual required")
    T = input.numel() // H
    check(_lib.lib().xllm_mi355_fused_add_rms_norm(_p(input), _p(residual), _p(weight), eps, T, H, H, _dt(input),
                                                  _stream()), "fused_add_rms_norm")


def fused_layernorm(input, weight, eps: float, residual: Optional[torch.Tensor] = None,
                    output: Optional[torch.Tensor] = None):
    """kernel::fused_layernorm CUDA/DCU branch (ops_api.cpp:364-372): rms_norm or fused_add_rms_norm."""
    if residual is not None:
        fused_add_rms_norm(input, residual, weight, eps)
        return input, residual
    out = torch.empty_like(input) if output is None else output
    rms_norm(out, input, weight, eps)
    return out, None


def rms_norm_static_fp8_quant(output, input, weight, scale, eps: float) -> None:
    """kernel::rms_norm_static_fp8_quant (ops_api.h:168, norm.cu:517-553)."""
    _need_cuda(output, input, weight, scale)
    H = input.size(-1)
    x2 = input.view(-1, H)
    check(_lib.lib().xllm_mi355_rms_norm_static_fp8_quant(_p(output), _p(x2), 0, _p(weight), _p(scale), eps, x2.size(0),
                                                         H, x2.stride(0), _dt(input), _stream()),
          "rms_norm_static_fp8_quant")


def fused_add_rms_norm_static_fp8_quant(output, input, residual, weight, scale, eps: float) -> None:
    """kernel::fused_add_rms_norm_static_fp8_quant (ops_api.h:172, norm.cu:555-...): residual updated in place."""
    _need_cuda(output, input, residual, weight, scale)
    H = input.size(-1)
    x2 = input.view(-1, H)
    check(_lib.lib().xllm_mi355_rms_norm_static_fp8_quant(_p(output), _p(x2), _p(residual), _p(weight), _p(scale), eps,
                                                         x2.size(0), H, x2.stride(0), _dt(input), _stream()),
          "fused_add_rms_norm_static_fp8_quant")


def rms_norm_dynamic_int8_quant(input, weight, eps: float, residual=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """N1 fusion (MLU fused_layernorm(dynamic_quant), param.h:243-277): (int8 [T,H], scale [T])."""
    _need_cuda(input, weight)
    H = input.size(-1)
    x2 = input.view(-1, H)
    q = torch.empty(x2.shape, dtype=torch.int8, device=input.device)
    s = torch.empty(x2.size(0), dtype=torch.float32, device=input.device)
    check(_lib.lib().xllm_mi355_rms_norm_dynamic_int8_quant(_p(q), _p(s), _p(x2), _p(residual), _p(weight), eps,
                                                           x2.size(0), H, x2.stride(0), _dt(input), _stream()),
          "rms_norm_dynamic_int8_quant")
    return q, s


# ------------------------------------------------------------------------------------------------ rope / act
def rotary_embedding(positions, query, key, cos_sin_cache, is_neox: bool = True, head_size: Optional[int] = None):
    """cuda::rotary_embedding(positions, q, k?, cos_sin_cache, is_neox) (cuda_ops_api.h:31-36, rope.cu:156-250).
    positions int64 [T]; q/k [T, n*head_size] views with a token stride; in place."""
    _need_cuda(positions, query, cos_sin_cache)
    if positions.dtype != torch.int64:
        raise Mi355Error("positions must be int64 (ops_api.cpp:118-195 converts before the call)")
    rot = cos_sin_cache.size(-1)
    hs = head_size if head_size is not None else (query.size(-1) if query.dim() == 3 else rot)
    T = positions.numel()
    q2 = query.reshape(T, -1) if query.dim() == 3 and query.is_contiguous() else query
    nq = q2.size(-1) // hs if q2.dim() == 2 else query.size(-2)
    nk = 0
    if key is not None:
        k2 = key.reshape(T, -1) if key.dim() == 3 and key.is_contiguous() else key
        nk = k2.size(-1) // hs if k2.dim() == 2 else key.size(-2)
    check(_lib.lib().xllm_mi355_rotary_embedding(
        _p(positions), _p(query), _p(key), _p(cos_sin_cache), T, nq, nk, hs, rot, query.stride(0),
        0 if key is None else key.stride(0), hs, int(is_neox), _dt(query), _stream()), "rotary_embedding")


def fused_qk_norm_rope(qkv, num_heads_q, num_heads_k, num_heads_v, head_dim, eps, q_weight, k_weight, cos_sin_cache,
                       interleaved, position_ids) -> None:
    """cuda::fused_qk_norm_rope (cuda_ops_api.h:235-249)."""
    _need_cuda(qkv, q_weight, k_weight, cos_sin_cache, position_ids)
    check(_lib.lib().xllm_mi355_fused_qk_norm_rope(
        _p(qkv), qkv.size(0), num_heads_q, num_heads_k, num_heads_v, head_dim, eps, _p(q_weight), _p(k_weight),
        _p(cos_sin_cache), _dt(cos_sin_cache), int(interleaved), _p(position_ids), _dt(qkv), _stream()),
        "fused_qk_norm_rope")


def act_and_mul(out, input, act_mode: str = "silu") -> None:
    """cuda::act_and_mul(out, in, mode) (cuda_ops_api.h:38-40, activation.cu:143-185)."""
    _need_cuda(out, input)
    if act_mode not in _ACT:
        raise Mi355Error(f"Unsupported act mode: {act_mode}")  # reference LOG(FATAL) activation.cu:182
    d = input.size(-1) // 2
    T = input.numel() // (2 * d)
    if not (input.is_contiguous() and out.is_contiguous()):
        raise Mi355Error("act_and_mul: contiguous tensors required")
    check(_lib.lib().xllm_mi355_act_and_mul(_p(out), _p(input), T, d, _ACT[act_mode], _dt(input), _stream()),
          "act_and_mul")


def act_and_mul_dynamic_int8_quant(input, act_mode: str = "silu", live_sizes=None):
    """N1 fusion (ScaledQuantizeParams.act_mode/is_gated, param.h:805-815). live_sizes (int32, device): the expert sizes of an
    expert-parallel rank -- only the first sum(live_sizes) sorted rows exist; a row past them is not read, gets scale 0 and keeps
    whatever bytes its quantised row held (the same on every kernel path)."""
    _need_cuda(input)
    d = input.size(-1) // 2
    T = input.numel() // (2 * d)
    q = torch.empty(T, d, dtype=torch.int8, device=input.device)
    s = torch.empty(T, dtype=torch.float32, device=input.device)
    if live_sizes is not None:
        _need_cuda(live_sizes)
        check(_lib.lib().xllm_mi355_act_and_mul_dynamic_int8_quant_live(
            _p(q), _p(s), _p(input), T, d, _ACT[act_mode], _dt(input), _p(live_sizes), live_sizes.numel(), _stream()),
            "act_and_mul_int8_live")
        return q, s
    check(_lib.lib().xllm_mi355_act_and_mul_dynamic_int8_quant(_p(q), _p(s), _p(input), T, d, _ACT[act_mode],
                                                              _dt(input), _stream()), "act_and_mul_int8")
    return q, s


# ------------------------------------------------------------------------------------------------ int8 W8A8
def scaled_quantize(x, output=None, output_scale=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """dcu::scaled_quantize(x, smooth=None, ...) -> (int8, fp32 scales) (dcu_ops_api.h, scaled_quantize.hip:411-...)."""
    _need_cuda(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise Mi355Error("scaled_quantize: x must be a contiguous [M, K] tensor")
    M, K = x.shape
    q = output if output is not None else torch.empty(M, K, dtype=torch.int8, device=x.device)
    s = output_scale if output_scale is not None else torch.empty(M, dtype=torch.float32, device=x.device)
    check(_lib.lib().xllm_mi355_scaled_quantize(_p(x), _p(q), _p(s), M, K, _dt(x), _stream()), "scaled_quantize")
    return q, s


_gemm_ws = {}
_GEMM_WS_BYTES = 64 << 20


def _ensure_gemm_workspace(device, nbytes=0):
    """the zero-at-rest split-K scratch of the row-major GEMM kernels: ONE fixed-size buffer per device, allocated on first
    use and never replaced (a captured HIP graph has its address baked in, and the int8 split-K path relies on it being zero
    between calls). Problems whose M*N*4 exceeds it simply do not split K (the C side checks ws_bytes). Returns its size."""
    ws = _gemm_ws.get(device)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("GEMM workspace must exist before a graph capture: run one eager step first")
        ws = torch.empty(_GEMM_WS_BYTES, dtype=torch.uint8, device=device)
        _gemm_ws[device] = ws
        check(_lib.lib().xllm_mi355_set_gemm_workspace(ws.data_ptr(), ws.numel()), "set_gemm_workspace")
        torch.cuda.synchronize(device)          # the zeroing memset runs on the null stream
    return ws.numel()


_slab_ws = {}
_SLAB_WS_BYTES = 64 << 20


def _slab_workspace(device):
    """explicit scratch of the packed-weight GEMMs (K-slice slabs; no invariant: nothing to zero, calls ordered on a stream may
    share it). One fixed buffer per device for the default stream and for captures, never replaced; every other stream that
    launches eagerly gets its own (lazily), and the dual micro-batch executor registers one per stream with
    set_gemm_workspace_for_stream."""
    cur = torch.cuda.current_stream(device)
    own = _slab_ws.get((device, cur.cuda_stream))
    if own is not None:
        return own
    if cur != torch.cuda.default_stream(device) and not torch.cuda.is_current_stream_capturing():
        # eager launches on a side stream (FusedMoE's shared experts next to the routed path) may run CONCURRENTLY with the main
        # stream's GEMMs: they get slabs of their own (a capture orders its launches explicitly and keeps the device buffer)
        own = torch.empty(_SLAB_WS_BYTES, dtype=torch.uint8, device=device)
        _slab_ws[(device, cur.cuda_stream)] = own
        return own
    ws = _slab_ws.get(device)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("GEMM slab workspace must exist before a graph capture: run one eager step first")
        ws = torch.empty(_SLAB_WS_BYTES, dtype=torch.uint8, device=device)
        _slab_ws[device] = ws
    return ws


_PACKED_POLICY = os.environ.get("XLLM_MI355_PACKED", "auto")   # "0" never, "1" wherever legal, "auto" = measured policy


def _prefer_packed(M: int, N: int, K: int) -> bool:
    """which int8 kernel serves a decode-shaped GEMM: since round 3 the weight-stream kernel on packed weights takes every
    problem of at most 512 rows (profiles/r03_gemm_ws.txt, r03_policy.txt). XLLM_MI355_PACKED = "0" / "1" = never /
    wherever legal (one switch for the int8, fp8 and 16-bit kinds); the round-2 policy arm left in round 4."""
    if _PACKED_POLICY == "0":
        return False
    if _PACKED_POLICY == "1":
        return M <= 512
    # round 3 (coalesced epilogue, eight-wave tile; profiles/r03_gemm_ws.txt, r03_policy.txt): the packed kernel serves every
    # decode-shaped problem. Stand-alone the two small projections are 3 us slower on it at M = 256 (22.3 / 19.1 us with their
    # slab pass against 19.4 / 17.9), but in the step their slabs feed the fused consumers (RoPE + KV write, add + norm + quant)
    # that run anyway: whole step 14.93 ms (round-2 policy) -> 14.85 (wide problems only) -> 14.51 ms (everything packed)
    return M <= 512


def pack_weight_i8(w: torch.Tensor) -> Optional[torch.Tensor]:
    """xllm_mi355_pack_weight_i8: [N, K] int8 row-major -> MFMA-fragment order for the weight-stream decode GEMM (done once,
    at weight-load time). None when the shape is outside the packed kernel's envelope (N % 16, K % 128)."""
    _need_cuda(w)
    N, K = w.shape
    if w.dtype != torch.int8 or not w.is_contiguous() or N % 16 or K % 128:
        return None
    out = torch.empty_like(w)
    check(_lib.lib().xllm_mi355_pack_weight_i8(_p(w), _p(out), N, K, _stream()), "pack_weight_i8")
    return out


_stream_ws = {}


def set_gemm_workspace_for_stream(stream: "torch.cuda.Stream", nbytes: int) -> None:
    """register a private split-K workspace for GEMMs launched on `stream` (micro-batches on concurrent streams)"""
    ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=stream.device)
    _stream_ws[stream.cuda_stream] = ws
    _slab_ws[(stream.device, stream.cuda_stream)] = torch.empty(_SLAB_WS_BYTES, dtype=torch.uint8, device=stream.device)
    _private_streams.add(stream.cuda_stream)
    check(_lib.lib().xllm_mi355_set_gemm_workspace_for_stream(stream.cuda_stream, ws.data_ptr(), ws.numel()),
          "set_gemm_workspace_for_stream")     # (the C side zeroes it: the int8 split-K scratch is zero at rest)
    torch.cuda.synchronize(stream.device)       # that memset runs on the null stream; side streams do not wait for it


def release_stream_workspaces(stream: "torch.cuda.Stream") -> None:
    """undo set_gemm_workspace_for_stream (the C side has 8 slots; stream handles are recycled by torch's pool)"""
    _stream_ws.pop(stream.cuda_stream, None)
    _slab_ws.pop((stream.device, stream.cuda_stream), None)
    _attn_ws.pop((stream.device, stream.cuda_stream), None)
    _private_streams.discard(stream.cuda_stream)
    check(_lib.lib().xllm_mi355_set_gemm_workspace_for_stream(stream.cuda_stream, 0, 0), "release gemm workspace")


def add_(a, b, out=None):
    """out = r16(a + b) for contiguous 16-bit tensors of one shape (xllm_mi355_add16; out may be a or b)"""
    _need_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float16) \
            or not a.is_contiguous() or not b.is_contiguous():
        raise Mi355Error("add_: two contiguous 16-bit tensors of one shape and dtype")
    out = out if out is not None else torch.empty_like(a)
    check(_lib.lib().xllm_mi355_add16(_p(out), _p(a), _p(b), a.numel(), _DT[a.dtype], _stream()), "add16")
    return out


def scaled_matmul(a, b, a_scale, b_scale, output_dtype=torch.bfloat16, bias=None, output=None, acc_out=None,
                  quant_bit_size: int = 8, a_quant_bit_size: int = 8, b_packed=None, c=None, alpha: float = 1.0,
                  beta: float = 1.0):
    """dcu::scaled_matmul (dcu_ops_api.h, scaled_matmul.cpp:103-300): a [M,K] int8, b [N,K] int8,
    a_scale [M]/[M,1] f32, b_scale [N]/[N,1] f32, optional bias [N] (output dtype).
    c (ScaledMatmulParams::c, kernels/param.h:852-866, alpha = beta = 1 only): out = r16(r16(a @ b ...) + c), the 16-bit GEMM
    result then a 16-bit add -- in the GEMM epilogue where the 8-phase kernel takes the shape (xllm_mi355_scaled_matmul_add), as a
    second pass (ops.add_) elsewhere; `output` may be c itself (in-place residual update)."""
    if c is not None:
        if alpha != 1.0 or beta != 1.0:
            raise Mi355Error("scaled_matmul: c is supported with alpha = beta = 1 only")
        _need_cuda(a, b, a_scale, b_scale, c)
        M, K = a.shape
        N = b.size(0)
        if c.shape != (M, N) or c.dtype != output_dtype or not c.is_contiguous():
            raise Mi355Error("scaled_matmul: c must be a contiguous [M, N] tensor of the output dtype")
        out = output if output is not None else torch.empty(M, N, dtype=output_dtype, device=a.device)
        rc = -2
        if a.is_contiguous() and b.is_contiguous() and M > 0:
            _ensure_gemm_workspace(a.device)
            rc = _lib.lib().xllm_mi355_scaled_matmul_add(_p(a), _p(b), _p(a_scale.reshape(-1)), _p(b_scale.reshape(-1)), _p(bias),
                                                         _p(c), _p(out), M, N, K, _DT[output_dtype], _stream())
        if rc != -2:
            check(rc, "scaled_matmul_add")
            return out
        y = scaled_matmul(a, b, a_scale, b_scale, output_dtype, bias, None, None, quant_bit_size, a_quant_bit_size, b_packed)
        return add_(y, c, out)
    _need_cuda(a, b, a_scale, b_scale)
    if quant_bit_size != 8 or a_quant_bit_size != 8:
        raise Mi355Error("scaled_matmul only supports w8a8 quantization")  # scaled_matmul.cpp:120-121
    if output_dtype not in (torch.bfloat16, torch.float16):
        raise Mi355Error("output dtype must be half or bfloat16")  # :123-125
    if a.dim() != 2 or b.dim() != 2 or a.dtype != torch.int8 or b.dtype != torch.int8 or a.size(1) != b.size(1) \
            or not a.is_contiguous() or not b.is_contiguous():
        raise Mi355Error("scaled_matmul: a [M,K] int8, b [N,K] int8, contiguous")
    M, K = a.shape
    N = b.size(0)
    out = output if output is not None else torch.empty(M, N, dtype=output_dtype, device=a.device)
    if b_packed is not None and _prefer_packed(M, N, K):   # decode-shaped: the weight-stream kernel on the pre-packed weights
        ws = _slab_workspace(a.device)
        rc = _lib.lib().xllm_mi355_scaled_matmul_packed(_p(a), _p(b_packed), _p(a_scale.reshape(-1)), _p(b_scale.reshape(-1)),
                                                        _p(bias), _p(out), _p(acc_out), M, N, K, _DT[output_dtype],
                                                        ws.data_ptr(), ws.numel(), _stream())
        if rc not in (-2, -4):   # XM_ERR_UNSUPPORTED / XM_ERR_WORKSPACE: a decline, the row-major kernel serves the shape
            check(rc, "scaled_matmul_packed")
            return out
    if acc_out is None:
        _ensure_gemm_workspace(a.device)
    check(_lib.lib().xllm_mi355_scaled_matmul(_p(a), _p(b), _p(a_scale.reshape(-1)), _p(b_scale.reshape(-1)), _p(bias),
                                             _p(out), _p(acc_out), M, N, K, _DT[output_dtype], _stream()),
          "scaled_matmul")
    return out


_row_amax = {}
_row_amax_retired = []


def _scratch_key(device):
    """which scratch buffer a launch on the CURRENT stream uses (the keying of _slab_workspace): one per device for the default
    stream and for graph captures (a capture orders its launches explicitly), one of its own for every stream that launches
    eagerly next to it (their kernels may run concurrently) and for the streams registered as private (dual micro-batch executor)"""
    cur = torch.cuda.current_stream(device)
    side = cur != torch.cuda.default_stream(device) and not torch.cuda.is_current_stream_capturing()
    return (device, cur.cuda_stream if (side or cur.cuda_stream in _private_streams) else 0)


def _row_amax_scratch(device, M: int):
    """zero-at-rest |max| scratch of the gate_up fusion, keyed like _slab_workspace: one buffer per device for the default stream
    and for captures, one per eagerly-launching side stream (their launches may run concurrently). A buffer is NEVER freed: a
    captured HIP graph has its address baked in (atomicMax + re-zero on replay), so a buffer a larger M outgrows is retired -- kept
    alive, still zero at rest -- and a bigger one takes its place for the launches that follow (round-3 advisor finding)."""
    key = _scratch_key(device)
    amax = _row_amax.get(key)
    if amax is None or amax.numel() < M:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("row-amax scratch must exist before a graph capture: run one eager step of this shape first")
        if amax is not None:
            _row_amax_retired.append(amax)
        amax = torch.zeros(max(M, 16384), dtype=torch.float32, device=device)     # zero at rest (the quantising pass re-zeroes)
        _row_amax[key] = amax
    return amax


_GATE_UP_FUSION = os.environ.get("XLLM_MI355_GATE_UP_FUSION", "1") == "1"   # A/B switch of the fusion below


def scaled_matmul_silu_mul_quant(a, b, a_scale, b_scale, output_dtype=torch.bfloat16, bias=None, b_packed=None):
    """N1 fusion across the GEMM boundary (round 3): the W8A8 gate_up projection -> SiLU(gate) * up -> per-token int8 quant of
    DenseMLP (dense_mlp.cpp:97-116 + the scaled_quantize of down_proj, linear.cpp:481-507) in TWO launches: the GEMM whose
    epilogue writes act [M, I] and the rows' |max| (xllm_mi355_scaled_matmul_gate_up_act), then one quantising pass
    (xllm_mi355_quantize_with_row_amax). a [M, K] int8, b [2 I, K] int8 (gate rows, then up rows). Returns (q [M, I] int8,
    scale [M]) -- bit-identical to scaled_matmul -> act_and_mul_dynamic_int8_quant -- or None outside the envelope."""
    _need_cuda(a, b, a_scale, b_scale)
    M, K = a.shape
    N = b.size(0)
    if not _GATE_UP_FUSION or N % 256 or K % 128 or not a.is_contiguous() or not b.is_contiguous() or M == 0:
        return None
    amax = _row_amax_scratch(a.device, M)
    I = N // 2
    act = torch.empty(M, I, dtype=output_dtype, device=a.device)
    rc = _lib.lib().xllm_mi355_scaled_matmul_gate_up_act(_p(a), _p(b), _p(b_packed), _p(a_scale.reshape(-1)),
                                                        _p(b_scale.reshape(-1)), _p(bias), _p(act), _p(amax), M, N, K,
                                                        _DT[output_dtype], 0, 0, _stream())
    if rc == -2:
        return None
    check(rc, "scaled_matmul_gate_up_act")
    q = torch.empty(M, I, dtype=torch.int8, device=a.device)
    qs = torch.empty(M, dtype=torch.float32, device=a.device)
    check(_lib.lib().xllm_mi355_quantize_with_row_amax(_p(act), _p(amax), _p(q), _p(qs), M, I, _DT[output_dtype], _stream()),
          "quantize_with_row_amax")
    return q, qs


# ------------------------------------------------------------------------------------------------ fp8
def scaled_matmul_add_rms_norm(a, b, a_scale, b_scale, residual, norm_weight, eps: float, bias=None,
                               quantize: bool = True, b_packed=None):
    """N1 fusion across the GEMM boundary: scaled_matmul (dcu::scaled_matmul) -> residual add + RMSNorm
    (kernel::fused_layernorm) [-> scaled_quantize]. `residual` [M, N] is updated in place to r16(y + residual);
    returns (q int8 [M, N], scale [M]) when `quantize`, else the 16-bit norm [M, N]. Bit-identical to the separate
    operators; returns None when the shape is outside the fused path's envelope (caller falls back)."""
    _need_cuda(a, b, a_scale, b_scale, residual, norm_weight)
    M, K = a.shape
    N = b.size(0)
    if not (a.is_contiguous() and b.is_contiguous() and residual.is_contiguous() and residual.shape == (M, N)):
        raise Mi355Error("scaled_matmul_add_rms_norm: contiguous a [M,K], b [N,K], residual [M,N]")
    if quantize:
        q = torch.empty(M, N, dtype=torch.int8, device=a.device)
        qs = torch.empty(M, dtype=torch.float32, device=a.device)
        out = None
    else:
        q = qs = None
        out = torch.empty(M, N, dtype=residual.dtype, device=a.device)
    rc = -2
    if b_packed is not None and _prefer_packed(M, N, K):
        ws = _slab_workspace(a.device)
        rc = _lib.lib().xllm_mi355_scaled_matmul_add_rms_norm_packed(
            _p(a), _p(b_packed), _p(a_scale.reshape(-1)), _p(b_scale.reshape(-1)), _p(bias), _p(residual), _p(norm_weight),
            eps, _p(out), _p(q), _p(qs), M, N, K, _dt(residual), ws.data_ptr(), ws.numel(), _stream())
    if rc == -2:
        if _ensure_gemm_workspace(a.device) < M * N * 4:
            return None
        rc = _lib.lib().xllm_mi355_scaled_matmul_add_rms_norm(
            _p(a), _p(b), _p(a_scale.reshape(-1)), _p(b_scale.reshape(-1)), _p(bias), _p(residual), _p(norm_weight), eps,
            _p(out), _p(q), _p(qs), M, N, K, _dt(residual), _stream())
    if rc == -2:  # XM_ERR_UNSUPPORTED: not a decode-shaped problem
        return None
    check(rc, "scaled_matmul_add_rms_norm")
    return (q, qs) if quantize else out


def static_scaled_fp8_quant(output, input, scale) -> None:
    """cuda::static_scaled_fp8_quant(out, in, scale) (cuda_ops_api.h:184-186, fp8_quant.cu:115-155)."""
    _need_cuda(output, input, scale)
    input_c = input.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_static_scaled_fp8_quant(_p(output), _p(input_c), _p(scale), input.numel(),
                                                       _dt(input), _stream()), "static_scaled_fp8_quant")


def fp8_scaled_quantize(input, output=None, scale=None):
    """cuda::fp8_scaled_quantize(in, out?, scale?) -> (q, scale) (fp8_scaled_quantize.cpp:20-50)."""
    _need_cuda(input)
    q = output if output is not None else torch.empty(input.shape, dtype=FP8, device=input.device)
    x = input.contiguous()
    if scale is not None:
        check(_lib.lib().xllm_mi355_fp8_scaled_quantize(_p(q), _p(x), _p(scale), 0, x.numel(), _dt(x), _stream()),
              "fp8_scaled_quantize")
        return q, scale
    s = torch.empty(1, dtype=torch.float32, device=input.device)
    # dynamic per-tensor scale: two launches (per-block maxima in a transient, stream-private scratch -> fold + quantise) instead of
    # the four graph nodes of the workspace-free entry (memset, amax with atomics, scale, quantise); same bits
    try:
        ws = _attn_workspace(input.device, 4096)
    except Mi355Error:      # first call inside a graph capture (no scratch yet, none may be allocated now): the workspace-free entry
        ws = None
    if ws is None:
        check(_lib.lib().xllm_mi355_fp8_scaled_quantize(_p(q), _p(x), 0, _p(s), x.numel(), _dt(x), _stream()),
              "fp8_scaled_quantize")
        return q, s
    check(_lib.lib().xllm_mi355_fp8_scaled_quantize_ws(_p(q), _p(x), _p(s), x.numel(), _dt(x), ws.data_ptr(), ws.numel(),
                                                       _stream()), "fp8_scaled_quantize")
    return q, s


def pack_weight_fp8(w: torch.Tensor) -> Optional[torch.Tensor]:
    """xllm_mi355_pack_weight_fp8: [N, K] e4m3 row-major -> MFMA-fragment order (the int8 byte permutation) for the weight-stream
    decode GEMM; done once at weight-load time. None outside the packed kernel's envelope (N % 16, K % 128)."""
    _need_cuda(w)
    N, K = w.shape
    if w.element_size() != 1 or not w.is_contiguous() or N % 16 or K % 128:
        return None
    out = torch.empty_like(w)
    check(_lib.lib().xllm_mi355_pack_weight_fp8(_p(w), _p(out), N, K, _stream()), "pack_weight_fp8")
    return out


_PACKED_FP8_POLICY = os.environ.get("XLLM_MI355_PACKED", "auto")   # "0" never, "1" wherever legal (M <= 512), "auto"


def _prefer_packed_fp8(M: int, N: int, K: int) -> bool:
    """decode-shaped fp8 GEMMs are one pass over the weights: the weight-stream kernel on packed weights (profiles/r03_gemm_ws.txt)"""
    if _PACKED_FP8_POLICY == "0":
        return False
    if _PACKED_FP8_POLICY == "1":
        return M <= 512
    return M <= 128


def fp8_scaled_matmul(a, b, a_scale, b_scale, output_dtype=torch.bfloat16, bias=None, output=None, b_packed=None):
    """cuda::fp8_scaled_matmul(a, b, a_scale, b_scale, out_dtype, bias?, out?) (fp8_scaled_matmul.cpp:20-47);
    a [M,K] e4m3fn, b [N,K] e4m3fn (the reference takes the [N,K] weight and forms b.t() itself). `b_packed` = pack_weight_fp8(b):
    decode-shaped calls then stream the weights in fragment order (same semantics, fp32 summation order differs)."""
    _need_cuda(a, b, a_scale, b_scale)
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(1):
        raise Mi355Error("fp8_scaled_matmul: a [M,K], b [N,K]")
    M, K = a.shape
    N = b.size(0)
    if a_scale.numel() not in (1, M) or b_scale.numel() not in (1, N):
        raise Mi355Error("fp8_scaled_matmul: scales must be scalar or per-token / per-channel")
    out = output if output is not None else torch.empty(M, N, dtype=output_dtype, device=a.device)
    if b_packed is not None and a.is_contiguous() and _prefer_packed_fp8(M, N, K):
        ws = _slab_workspace(a.device)
        rc = _lib.lib().xllm_mi355_fp8_scaled_matmul_packed(_p(a), _p(b_packed), _p(a_scale), a_scale.numel(), _p(b_scale),
                                                            b_scale.numel(), _p(bias), _p(out), M, N, K, _DT[output_dtype],
                                                            ws.data_ptr(), ws.numel(), _stream())
        if rc not in (-2, -4):   # XM_ERR_UNSUPPORTED / XM_ERR_WORKSPACE: the row-major kernel serves the shape
            check(rc, "fp8_scaled_matmul_packed")
            return out
    if M <= 512:
        _ensure_gemm_workspace(a.device, 1)  # decode shapes may split K through it (fp32 slabs, deterministic reduce)
    check(_lib.lib().xllm_mi355_fp8_scaled_matmul(_p(a), _p(b), _p(a_scale), a_scale.numel(), _p(b_scale),
                                                 b_scale.numel(), _p(bias), _p(out), M, N, K, _DT[output_dtype],
                                                 _stream()), "fp8_scaled_matmul")
    return out


def pack_weight_16(w: torch.Tensor) -> Optional[torch.Tensor]:
    """xllm_mi355_pack_weight_16: [N, K] bf16 / f16 row-major -> MFMA-fragment order (the 8-bit kinds' byte permutation on rows of
    2 K bytes) for the weight-stream decode GEMM; once, at weight-load time. None outside the envelope (N % 16, K % 64)."""
    _need_cuda(w)
    N, K = w.shape
    if w.dtype not in (torch.bfloat16, torch.float16) or not w.is_contiguous() or N % 16 or K % 64:
        return None
    out = torch.empty_like(w)
    check(_lib.lib().xllm_mi355_pack_weight_16(_p(w), _p(out), N, K, _stream()), "pack_weight_16")
    return out


_PACKED_16_POLICY = os.environ.get("XLLM_MI355_PACKED", "auto")   # "0" never, "1" wherever legal (M <= 512), "auto"


def _prefer_packed_16(M: int, N: int, K: int) -> bool:
    """16-bit decode linears on packed weights: a single pass over the weights in fragment order"""
    if _PACKED_16_POLICY == "0":
        return False
    if _PACKED_16_POLICY == "1":
        return M <= 512
    # measured (profiles/r03_gemm_ws16.txt, Qwen2-7B shapes, us): M = 64 gate_up 79.6 -> 46.1 (6.0 TB/s), down 45.2 -> 32.9,
    # lm_head 257 -> 218; M = 256 gate_up 94.7 -> 79.8, down 79.4 -> 51.7; at M <= 32 the small projections (qkv, o) are 1-2 us
    # faster on the no-LDS row-major weight-stream kernel (gemm_wsb.hip), the large ones 15-23 us faster here
    return M <= 512 and (M > 32 or N * K >= 48 * 1024 * 1024)


def matmul(a, b, bias=None, b_packed=None):
    """dcu::matmul(a, w, bias) == F::linear (kernels/dcu/matmul.cpp:20-25); a [..., K], w [N, K]. `b_packed` = pack_weight_16(w):
    decode-shaped calls then stream the weights in fragment order (same semantics, another fp32 summation order)."""
    _need_cuda(a, b)
    K = a.size(-1)
    a2 = a.reshape(-1, K)
    if not a2.is_contiguous():
        a2 = a2.contiguous()
    N = b.size(0)
    out = torch.empty(a2.size(0), N, dtype=a.dtype, device=a.device)
    if b_packed is not None and _prefer_packed_16(a2.size(0), N, K):
        ws = _slab_workspace(a.device)
        rc = _lib.lib().xllm_mi355_matmul_packed(_p(a2), _p(b_packed), _p(bias), _p(out), a2.size(0), N, K, _dt(a),
                                                 ws.data_ptr(), ws.numel(), _stream())
        if rc not in (-2, -4):
            check(rc, "matmul_packed")
            return out.view(*a.shape[:-1], N)
    if a2.size(0) <= 512 or ((a2.size(0) + 127) // 128) * ((N + 127) // 128) < 128:
        # decode shapes -- and tall problems with few columns (router gates) -- may split K through it (fp32 slabs, deterministic reduce)
        _ensure_gemm_workspace(a.device, 1)
    b_c = b.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_matmul(_p(a2), _p(b_c), _p(bias), _p(out), a2.size(0), N, K, _dt(a),
                                      _stream()), "matmul")
    return out.view(*a.shape[:-1], N)


def matmul_silu_mul(a, b, bias=None, b_packed=None):
    """N1 fusion across the GEMM boundary for an UNQUANTISED dense MLP (dense_mlp.cpp:97-116: gate_up_proj -> act_and_mul): the
    16-bit gate_up projection on packed weights with SiLU(gate) * up in its epilogue (xllm_mi355_matmul_gate_up_act). a [M, K],
    b [2 I, K] (gate rows, then up rows), b_packed = pack_weight_16(b). Returns act [M, I] -- the expression of matmul ->
    act_and_mul on the packed kernel's fp32 sums (their order is the tile plan's, so the two agree to fp32 rounding) -- or None
    outside the envelope (caller runs the two operators)."""
    _need_cuda(a, b)
    M, K = a.shape
    N = b.size(0)
    if not _GATE_UP_FUSION or b_packed is None or N % 32 or M == 0 or not a.is_contiguous() or not _prefer_packed_16(M, N, K):
        return None
    act = torch.empty(M, N // 2, dtype=a.dtype, device=a.device)
    rc = _lib.lib().xllm_mi355_matmul_gate_up_act(_p(a), _p(b_packed), _p(bias), _p(act), M, N, K, _dt(a), 0, 0, _stream())
    if rc in (-2, -4):
        return None
    check(rc, "matmul_gate_up_act")
    return act


_argmax_ws = {}
# lm_head + greedy argmax in one pass: "auto" = under tensor parallelism only (it replaces the [B, V / tp] logits all-gather by [B]
# pairs; on one GPU it ties with the two operators: profiles/r04_lm_head_ab.txt), "1" always, "0" never
_GREEDY_FUSION = os.environ.get("XLLM_MI355_GREEDY_FUSION", "auto")


def matmul_argmax(a, b_packed, N: int, bias=None, want_value: bool = False):
    """lm_head + Sampler::greedy_sample in one pass (xllm_mi355_matmul_argmax_packed, round 4): token ids int64 [M] =
    argmax_n r16(a . w[n] + bias[n]) with torch.argmax's order, the [M, N] logits never written. a [M, K] 16-bit contiguous,
    b_packed = pack_weight_16(w [N, K]). want_value: also the winning logit (float32 [M]) -- what a column-sharded lm_head exchanges
    instead of its logits. None outside the packed kernel's envelope (M > 512, N % 16, ...): the caller runs matmul + greedy_argmax."""
    _need_cuda(a, b_packed)
    M, K = a.shape
    if b_packed is None or M == 0 or M > 512 or N % 16 or not a.is_contiguous():
        return None
    need = _lib.lib().xllm_mi355_matmul_argmax_workspace_bytes(M, N)
    key = _scratch_key(a.device)
    ws = _argmax_ws.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("matmul_argmax scratch must exist before a graph capture: run one eager step of this shape first")
        if ws is not None:
            _retired_ws.append(ws)           # a captured graph may still write it
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        _argmax_ws[key] = ws
    idx = torch.empty(M, dtype=torch.int64, device=a.device)
    val = torch.empty(M, dtype=torch.float32, device=a.device) if want_value else None
    rc = _lib.lib().xllm_mi355_matmul_argmax_packed(_p(a), _p(b_packed), _p(bias), _p(idx), _p(val), M, N, K, _dt(a), _p(ws),
                                                    ws.numel(), _stream())
    if rc in (-2, -4):
        return None
    check(rc, "matmul_argmax_packed")
    return (idx, val) if want_value else idx


# ------------------------------------------------------------------------------------------------ attention
_attn_ws = {}
_retired_ws = []   # outgrown scratch buffers stay allocated: a captured HIP graph may still launch kernels that write them
_private_streams = set()   # streams registered with set_gemm_workspace_for_stream: they get scratch buffers of their own


def _attn_workspace(device, nbytes):
    """split-KV partials: transient per launch. One buffer per device (launches ordered on a stream, or across streams that
    wait on each other, may share it); a stream registered as running CONCURRENTLY with others (set_gemm_workspace_for_stream:
    the dual micro-batch executor) gets its own. Grows by replacement outside captures only (the old buffer is retired, not
    freed, so a captured graph that still points at it stays valid)."""
    sid = torch.cuda.current_stream(device).cuda_stream
    key = (device, sid) if sid in _private_streams else device
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("attention workspace too small inside a graph capture: run one eager step of this shape first")
        if ws is not None:
            _retired_ws.append(ws)
        ws = torch.empty(max(nbytes, 16 << 20), dtype=torch.uint8, device=device)
        _attn_ws[key] = ws
    return ws


def prefill_attention(q, k, v, cu_seqlens_q, cu_seqlens_k, max_q_len, scale, is_causal=True, window_left=-1, out=None):
    """prefix_prefill_varlen_fwd argument set (layers/dcu/flash_attention.cpp:45-72, 167-218):
    q [Tq, nq, d], k/v [Tk, nkv, d] (token-strided views of qkv allowed) -> out [Tq, nq*d]."""
    _need_cuda(q, k, v, cu_seqlens_q, cu_seqlens_k)
    Tq, nq, d = q.shape
    nkv = k.size(1)
    o = out if out is not None else torch.empty(Tq, nq * d, dtype=q.dtype, device=q.device)
    check(_lib.lib().xllm_mi355_prefill_attention(
        _p(q), _p(k), _p(v), _p(o), _p(cu_seqlens_q), _p(cu_seqlens_k), cu_seqlens_q.numel() - 1, nq, nkv, d,
        q.stride(0), k.stride(0), v.stride(0), max_q_len, scale, int(is_causal), window_left, _dt(q), _stream()),
        "prefill_attention")
    return o


def paged_attention(q, k_cache, v_cache, cu_seqlens_q, kv_seq_lens, block_table, max_q_len, max_kv_len, scale,
                    is_causal=False, window_left=-1, out=None):
    """prefix_decode_varlen_fwd argument set (layers/dcu/flash_attention.cpp:74-94, 220-288):
    q [Tq, nq, d] packed by cu_seqlens_q (None => one token per sequence); caches [n_blocks, bs, nkv, d]."""
    _need_cuda(q, k_cache, v_cache, kv_seq_lens, block_table)
    Tq, nq, d = q.shape
    n_blocks, bs, nkv, _ = k_cache.shape
    B = kv_seq_lens.numel()
    o = out if out is not None else torch.empty(Tq, nq * d, dtype=q.dtype, device=q.device)
    need = _lib.lib().xllm_mi355_paged_attention_workspace_bytes(B, nq, d, max_q_len, Tq)
    ws = _attn_workspace(q.device, need)
    bt = block_table if block_table.is_contiguous() else block_table.contiguous()
    check(_lib.lib().xllm_mi355_paged_attention(
        _p(q), _p(k_cache), _p(v_cache), _p(o), _p(cu_seqlens_q), _p(kv_seq_lens), _p(bt), bt.size(1), B, Tq, nq, nkv,
        d, bs, n_blocks, q.stride(0), max_q_len, max_kv_len, scale, int(is_causal), window_left, _dt(q), _p(ws),
        ws.numel(), _stream()), "paged_attention")
    return o


def mla_decode(q, k_cache, seqlens_k, block_table, head_size_v: int, scale: float, max_kv_len: int, out=None):
    """flash_mla::dense_decode argument set (kernels/dcu/flash_mla_adapter.h:40-50): q [B, H, 576] =
    [q_nope*W_kc || q_pe], k_cache [n_blocks, block, 1, 576]; returns [B, H, head_size_v]."""
    _need_cuda(q, k_cache, seqlens_k, block_table)
    B, H, D = q.shape
    n_blocks, bs = k_cache.shape[0], k_cache.shape[1]
    o = out if out is not None else torch.empty(B, H, head_size_v, dtype=q.dtype, device=q.device)
    ws = _attn_workspace(q.device, B * H * 32 * (head_size_v + 2) * 4)
    bt = block_table if block_table.is_contiguous() else block_table.contiguous()
    q_c = q.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_mla_decode(_p(q_c), _p(k_cache), _p(o), _p(seqlens_k), _p(bt), bt.size(1), B, H,
                                          D, head_size_v, bs, n_blocks, max_kv_len, scale, _dt(q), _p(ws), ws.numel(),
                                          _stream()), "mla_decode")
    return o


# ------------------------------------------------------------------------------------------------ MoE
_moe_ws = {}


def mla_prefill(q, k_cache, cu_seqlens_q, kv_seq_lens, block_table, head_size_v: int, scale: float, max_kv_len: int,
                is_causal: bool = True, out=None):
    """DeepseekV2AttentionImpl::prefill_sdpa (layers/dcu/deepseek_v2_attention.cpp:212-262) over the paged latent
    cache: q [T, H, 576], out [T, H, head_size_v]; bottom-right causal alignment."""
    _need_cuda(q, k_cache, cu_seqlens_q, kv_seq_lens, block_table)
    T, H, D = q.shape
    if not q.is_contiguous() or not block_table.is_contiguous():
        raise Mi355Error("q and block_table must be contiguous")
    out = out if out is not None else torch.empty(T, H, head_size_v, dtype=q.dtype, device=q.device)
    ws_bytes = ((T * 8 + 255) // 256) * 256 + 8 * T * H * (head_size_v + 2) * 4
    ws = _attn_workspace(q.device, ws_bytes)
    check(_lib.lib().xllm_mi355_mla_prefill(
        _p(q), _p(k_cache), _p(out), _p(cu_seqlens_q), _p(kv_seq_lens), _p(block_table), block_table.size(1),
        kv_seq_lens.numel(), T, H, D, head_size_v, k_cache.size(1), k_cache.size(0), max_kv_len, scale,
        int(is_causal), _dt(q), _p(ws), ws.numel() * ws.element_size(), _stream()), "mla_prefill")
    return out


def moe_compute_index(expert_id, num_experts: int):
    """kernel::moe_gen_idx (ops_api.h:73) -> cuda::moe_compute_index (moe/moe_compute_index.cu:111-160):
    expert_id [T, topk] int32 -> (src_dst [T*topk], dst_src [T*topk], expert_sizes [E]); stable order."""
    _need_cuda(expert_id)
    T, topk = expert_id.shape
    dev = expert_id.device
    need = 4 * ((T * topk + 1023) // 1024 + 1) * num_experts
    ws = _moe_ws.get(dev)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error("MoE workspace too small inside a graph capture: run one eager step of this shape first")
        if ws is not None:
            _retired_ws.append(ws)
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=dev)
        _moe_ws[dev] = ws
        check(_lib.lib().xllm_mi355_set_moe_workspace(ws.data_ptr(), ws.numel()), "set_moe_workspace")
    src_dst = torch.empty(T * topk, dtype=torch.int32, device=dev)
    dst_src = torch.empty(T * topk, dtype=torch.int32, device=dev)
    sizes = torch.empty(num_experts, dtype=torch.int32, device=dev)
    expert_id_c = expert_id.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_moe_compute_index(_p(expert_id_c), T, topk, num_experts, _p(src_dst),
                                                 _p(dst_src), _p(sizes), _stream()), "moe_compute_index")
    return src_dst, dst_src, sizes


def moe_combine_result(gemm2, weights, n_tokens: int, topk: int):
    """kernel::moe_combine_result (ops_api.h:77) -> moe/moe_combine.cu:38-62"""
    _need_cuda(gemm2, weights)
    H = gemm2.size(-1)
    out = torch.empty(n_tokens, H, dtype=gemm2.dtype, device=gemm2.device)
    gemm2_c = gemm2.contiguous()  # keep the (possibly new) tensor alive across the call
    weights_c = weights.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_moe_combine(_p(out), _p(gemm2_c), _p(weights_c), n_tokens, topk, H,
                                           _dt(gemm2), _stream()), "moe_combine_result")
    return out


def moe_combine_sorted(gemm2_sorted, src_dst, weights, n_tokens: int, topk: int, local_expert_sizes=None):
    """index_copy_ + kernel::moe_combine_result in one pass (layers/dcu/fused_moe.cpp:296-303): gemm2_sorted holds the rows
    of the second grouped GEMM in expert order; out[t] = sum_k w[t,k] * gemm2_sorted[src_dst[t*topk+k]].
    local_expert_sizes (int32 [E_local], device; EP rank): only the first sum(local_expert_sizes) sorted rows exist, the
    others count as the zero rows of the reference's gemm2_full and are skipped."""
    _need_cuda(gemm2_sorted, src_dst, weights)
    H = gemm2_sorted.size(-1)
    out = torch.empty(n_tokens, H, dtype=gemm2_sorted.dtype, device=gemm2_sorted.device)
    gemm2_c = gemm2_sorted.contiguous()  # keep the (possibly new) tensors alive across the call
    src_dst_c = src_dst.contiguous()
    weights_c = weights.contiguous()
    if local_expert_sizes is not None:
        sizes_c = local_expert_sizes.contiguous()
        check(_lib.lib().xllm_mi355_moe_combine_sorted_local(_p(out), _p(gemm2_c), _p(src_dst_c), _p(weights_c), _p(sizes_c),
                                                            sizes_c.numel(), n_tokens, topk, H, _dt(gemm2_sorted), _stream()),
              "moe_combine_sorted_local")
        return out
    check(_lib.lib().xllm_mi355_moe_combine_sorted(_p(out), _p(gemm2_c), _p(src_dst_c), _p(weights_c), n_tokens, topk, H,
                                                  _dt(gemm2_sorted), _stream()), "moe_combine_sorted")
    return out


def bmm_heads(x, w_nk, out=None) -> torch.Tensor:
    """MLA weight absorption without rocBLAS (round 5): out[t, h, n] = sum_k x[t, h, k] * w_nk[h, n, k] -- torch::bmm(x.transpose(0, 1),
    w).transpose(0, 1) of DeepseekV2AttentionImpl (layers/dcu/deepseek_v2_attention.cpp:180-187, 310-311) with the head's matrix
    stored K-contiguous ([h, N, K]: kv_b_proj's own slice for W_vc; W_kc transposed once at load). x [T, h, K] and out [T, h, N] may
    be strided views (x: a slice of the packed q tensor); no transpose, no copy. Raises when the shape is outside the kernel."""
    _need_cuda(x, w_nk)
    T, H, K = x.shape
    N = w_nk.size(1)
    if w_nk.size(0) != H or w_nk.size(2) != K or x.stride(2) != 1 or w_nk.stride(2) != 1 or x.dtype != w_nk.dtype:
        raise Mi355Error("bmm_heads: x [T, h, K] and w [h, N, K] with K contiguous and one dtype")
    if out is None:
        out = torch.empty(T, H, N, dtype=x.dtype, device=x.device)
    if out.stride(2) != 1:
        raise Mi355Error("bmm_heads: out must be contiguous over N")
    check(_lib.lib().xllm_mi355_bmm_heads(_p(x), x.stride(0), x.stride(1), _p(w_nk), w_nk.stride(0), w_nk.stride(1), _p(out),
                                          out.stride(0), out.stride(1), T, H, N, K, _dt(x), _stream()), "bmm_heads")
    return out


def group_gemm(input, weight, token_count, output=None):
    """dcu::group_gemm(input [total, K], weight [E, N, K], token_count [E] int32 (device), out?) (dcu_ops_api.h:48-51)"""
    _need_cuda(input, weight, token_count)
    E, N, K = weight.shape
    out = output if output is not None else torch.empty(input.size(0), N, dtype=input.dtype, device=input.device)
    input_c = input.contiguous()  # keep the (possibly new) tensor alive across the call
    weight_c = weight.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_group_gemm(_p(input_c), _p(weight_c), _p(token_count), _p(out),
                                          input.size(0), E, N, K, _dt(input), _stream()), "group_gemm")
    return out


def group_gemm_gather(input, row_index, index_div: int, weight, token_count):
    """index_select(input, row_index / index_div) + dcu::group_gemm without the expanded copy (layers/dcu/fused_moe.cpp:195-197,
    250-262); returns None when the 256x256 kernel cannot take the shape (the caller then expands and calls group_gemm)"""
    _need_cuda(input, row_index, weight, token_count)
    E, N, K = weight.shape
    rows = row_index.numel()
    out = torch.empty(rows, N, dtype=input.dtype, device=input.device)
    input_c = input.contiguous()  # keep the (possibly new) tensors alive across the call
    weight_c = weight.contiguous()
    index_c = row_index.contiguous()
    rc = _lib.lib().xllm_mi355_group_gemm_gather(_p(input_c), input.size(0), _p(index_c), index_div, _p(weight_c),
                                                 _p(token_count), _p(out), rows, E, N, K, _dt(input), _stream())
    if rc == -2:  # XM_ERR_UNSUPPORTED
        return None
    check(rc, "group_gemm_gather")
    return out


def group_gemm_w8a8(a, a_scale, weight, w_scale, token_count, output_dtype=torch.bfloat16, row_index=None, index_div: int = 1):
    """W8A8 grouped GEMM (GroupGemmParams.a_scale / b_scale, param.h:374-394): a int8 [rows, K] sorted by expert with
    a_scale [rows] -- or, with row_index, the un-expanded activations [T, K] / [T] gathered as row_index[r] / index_div;
    weight int8 [E, N, K], w_scale float32 [E, N]; token_count int32 [E] (device). Returns 16-bit [rows, N]."""
    _need_cuda(a, a_scale, weight, w_scale, token_count)
    E, N, K = weight.shape
    rows = row_index.numel() if row_index is not None else a.size(0)
    need = 16 * (rows // 256 + E + 2) + 64
    ws = _moe_ws.get(a.device)
    if ws is not None and ws.numel() < need:
        _retired_ws.append(ws)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=a.device)
        _moe_ws[a.device] = ws
        check(_lib.lib().xllm_mi355_set_moe_workspace(ws.data_ptr(), ws.numel()), "set_moe_workspace")
    out = torch.empty(rows, N, dtype=output_dtype, device=a.device)
    a_c, as_c, w_c = a.contiguous(), a_scale.contiguous(), weight.contiguous()  # keep the (possibly new) tensors alive
    ws_c = w_scale.to(torch.float32).contiguous()
    idx_c = row_index.contiguous() if row_index is not None else None
    check(_lib.lib().xllm_mi355_group_gemm_w8a8(_p(a_c), a.size(0), _p(as_c), _p(idx_c), index_div, _p(w_c), _p(ws_c),
                                                _p(token_count), _p(out), rows, E, N, K, _DT[output_dtype], _stream()),
          "group_gemm_w8a8")
    return out


# ------------------------------------------------------------------------------------------------ N1 fusions
def rotary_embedding_and_cache(positions, query, key, value, cos_sin_cache, slot_ids, key_cache, value_cache,
                               head_size: int, is_neox: bool = True) -> None:
    """apply_rotary + reshape_paged_cache in one launch (bit-identical to the two operators)."""
    _need_cuda(positions, query, key, value, cos_sin_cache, slot_ids, key_cache, value_cache)
    T = positions.numel()
    nq, nk = query.size(-1) // head_size, key.size(-1) // head_size
    check(_lib.lib().xllm_mi355_rotary_embedding_and_cache(
        _p(positions), _p(query), _p(key), _p(value), _p(cos_sin_cache), _p(slot_ids), _p(key_cache), _p(value_cache),
        T, nq, nk, head_size, cos_sin_cache.size(-1), query.stride(0), key.stride(0), value.stride(0),
        key_cache.size(-3), key_cache.size(0), int(is_neox), _dt(query), _stream()), "rotary_embedding_and_cache")


_QKV_ROPE_FUSION = os.environ.get("XLLM_MI355_QKV_ROPE", "1") != "0"


def scaled_matmul_rope_cache(a, b_packed, a_scale, b_scale, bias, positions, cos_sin_cache, slot_ids, key_cache, value_cache,
                             n_q_heads: int, n_kv_heads: int, head_size: int, output_dtype=torch.bfloat16, is_neox: bool = True):
    """N1 fusion across the GEMM boundary: W8A8 qkv projection on packed weights -> dequant epilogue -> RoPE -> KV write
    (scaled_matmul + rotary_embedding_and_cache, bit for bit). Returns the packed qkv rows [M, N] with q and k rotated, or None
    when the shape is outside the fused path (the caller runs the separate operators)."""
    _need_cuda(a, b_packed, a_scale, b_scale, positions, cos_sin_cache, slot_ids, key_cache, value_cache)
    M, K = a.shape
    N = (n_q_heads + 2 * n_kv_heads) * head_size
    if not _QKV_ROPE_FUSION or b_packed.numel() != N * K or not _prefer_packed(M, N, K) or cos_sin_cache.dtype != output_dtype:
        return None
    ws = _slab_workspace(a.device)
    out = torch.empty(M, N, dtype=output_dtype, device=a.device)
    rc = _lib.lib().xllm_mi355_scaled_matmul_rope_cache_packed(
        _p(a), _p(b_packed), _p(a_scale), _p(b_scale), _p(bias), _p(out), M, N, K, _DT[output_dtype], _p(positions),
        _p(cos_sin_cache), _p(slot_ids), _p(key_cache), _p(value_cache), n_q_heads, n_kv_heads, head_size,
        cos_sin_cache.size(-1), key_cache.size(-3), key_cache.size(0), int(is_neox), _p(ws), ws.numel(), _stream())
    if rc == -2:
        return None
    check(rc, "scaled_matmul_rope_cache")
    return out


# merge + int8 quantisation of split-KV partials in ONE finishing launch (bit-identical to paged_attention + scaled_quantize).
# Default since round 6: with every load of the launch in flight at once (and 1024 threads per token for rows beyond 1024
# elements) it takes 4.7 us against 4.7 + 4.6 us for the merge and quantise launches it replaces (round 2's form waited for the
# (m, l) pairs and then for each split in turn: 12.7 us, opt-in only); XLLM_MI355_ATTN_FINISH=0 = the two operators
_ATTN_FINISH = os.environ.get("XLLM_MI355_ATTN_FINISH", "1") == "1"


def paged_decode_attention_int8(q, k_cache, v_cache, kv_seq_lens, block_table, max_kv_len, scale, window_left=-1,
                                want_16bit: bool = False):
    """decode attention whose epilogue also emits scaled_quantize of its output: returns (int8 [B, nq*d], scale [B],
    16-bit out or None), or None when the shape needs split-KV (caller falls back to paged_attention + quant)."""
    _need_cuda(q, k_cache, v_cache, kv_seq_lens, block_table)
    B, nq, d = q.shape
    n_blocks, bs, nkv, _ = k_cache.shape
    oq = torch.empty(B, nq * d, dtype=torch.int8, device=q.device)
    os_ = torch.empty(B, dtype=torch.float32, device=q.device)
    o16 = torch.empty(B, nq * d, dtype=q.dtype, device=q.device) if want_16bit else None
    bt = block_table if block_table.is_contiguous() else block_table.contiguous()
    # plans that split the token range over the grid finish through ONE merge + quantise launch
    # (XLLM_MI355_ATTN_FINISH=0: decline, the caller runs paged_attention + scaled_quantize)
    ws = None
    if _ATTN_FINISH:
        ws = _attn_workspace(q.device, _lib.lib().xllm_mi355_paged_attention_workspace_bytes(B, nq, d, 1, B))
    rc = _lib.lib().xllm_mi355_paged_decode_attention_int8_ws(
        _p(q), _p(k_cache), _p(v_cache), _p(o16), _p(oq), _p(os_), _p(kv_seq_lens), _p(bt), bt.size(1), B, nq, nkv, d,
        bs, q.stride(0), max_kv_len, scale, window_left, _dt(q), _p(ws), ws.numel() if ws is not None else 0, _stream())
    if rc == -2:  # XM_ERR_UNSUPPORTED: split-KV shape without a workspace
        return None
    check(rc, "paged_decode_attention_int8")
    return oq, os_, o16


def decode_metadata_update(src: dict, dst: dict, actual_num_tokens: int, padded_num_tokens: int,
                           actual_batch_size: int, actual_indices_size: int, padded_batch_size: int = 0) -> None:
    """N2: refresh the persistent decode metadata of a captured graph on the device
    (cuda::update_llm_decode_metadata, kernels/cuda/llm_decode_metadata_update.cu:27-60) and, when dst holds
    "block_table" / "kv_lens", rebuild the dense 0-padded block table + per-sequence lengths in the same launch.
    src keys: tokens positions new_cache_slots kv_seq_lens paged_kv_indptr paged_kv_indices paged_kv_last_page_len;
    dst keys: the same + kv_seq_lens_delta, block_table [padded_B, max_blocks], kv_lens [padded_B]."""
    md = _lib.DecodeMetadata()
    for k in ("tokens", "positions", "new_cache_slots", "kv_seq_lens", "paged_kv_indptr", "paged_kv_indices",
              "paged_kv_last_page_len"):
        t = src.get(k)
        if t is not None:
            _need_cuda(t)
            assert t.dtype == torch.int32 and t.is_contiguous()
        setattr(md, "src_" + k, _p(t) or None)
    for k in ("tokens", "positions", "new_cache_slots", "kv_seq_lens", "kv_seq_lens_delta", "paged_kv_indptr",
              "paged_kv_indices", "paged_kv_last_page_len", "block_table", "kv_lens"):
        t = dst.get(k)
        if t is not None:
            _need_cuda(t)
            assert t.dtype == torch.int32 and t.is_contiguous()
        setattr(md, "dst_" + k, _p(t) or None)
    md.actual_num_tokens, md.padded_num_tokens = actual_num_tokens, padded_num_tokens
    md.actual_batch_size, md.actual_indices_size = actual_batch_size, actual_indices_size
    bt = dst.get("block_table")
    md.max_blocks_per_seq = bt.size(1) if bt is not None else 0
    md.padded_batch_size = padded_batch_size or (bt.size(0) if bt is not None else actual_batch_size)
    check(_lib.lib().xllm_mi355_decode_metadata_update(C.byref(md), _stream()), "decode_metadata_update")


# ------------------------------------------------------------------------------------------------ N3: sampler
def philox_uniform(n: int, seed: int, offset: int, device="cuda", out=None):
    """u[i] = the first hiprand_uniform() of hiprand_init(seed, subsequence=i, offset) (Philox4x32-10); `out` (float32 [n],
    contiguous) is filled in place when given"""
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=device)
    elif out.dtype != torch.float32 or out.numel() != n or not out.is_contiguous() or not out.is_cuda:
        raise Mi355Error("philox_uniform: out must be a contiguous float32 device tensor of n elements")
    check(_lib.lib().xllm_mi355_philox_uniform(_p(out), n, seed, offset, _stream()), "philox_uniform")
    return out


def random_sample(probs, uniform=None, seed: int = 0, offset: int = 0):
    """dcu::random_sample (kernels/dcu/random_sample.hip:240-270): probs [B, V] or [B, S, V] fp32 -> int32 ids"""
    _need_cuda(probs)
    if probs.dim() not in (2, 3) or probs.dtype != torch.float32:
        raise Mi355Error("probs must be a 2D or 3D float32 tensor")   # reference CHECKs :241-254
    flat = probs.reshape(-1, probs.size(-1)).contiguous()
    out = torch.empty(flat.size(0), dtype=torch.int32, device=probs.device)
    check(_lib.lib().xllm_mi355_random_sample(_p(flat), _p(out), flat.size(0), flat.size(1), _p(uniform), seed, offset,
                                              _stream()), "random_sample")
    return out.view(probs.shape[:-1])


def softmax_random_sample(logits, uniform=None, seed: int = 0, offset: int = 0, do_sample=None):
    """Sampler::forward's tail (sampler.cpp:118-137) in one launch: softmax(logits, -1, fp32) -> random_sample (rows with
    do_sample False: greedy = the first column of the maximum) WITHOUT materialising the [B, V] probabilities
    (xllm_mi355_softmax_random_sample). logits [B, V] fp32 / bf16 / f16, already temperature-scaled and top-k / top-p masked;
    returns int32 [B]"""
    B, V, stride = _logits_2d(logits)
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    ds = None
    if do_sample is not None:
        _need_cuda(do_sample)
        ds = do_sample.to(torch.uint8).contiguous()
    u = None if uniform is None else uniform.to(torch.float32).contiguous()
    check(_lib.lib().xllm_mi355_softmax_random_sample(_p(logits), _p(out), B, V, stride, _dt(logits), _p(u), seed, offset, _p(ds),
                                                      _stream()), "softmax_random_sample")
    return out


def sample_top_k_top_p(logits, temperatures=None, top_k=None, top_p=None, uniform=None, seed: int = 0, offset: int = 0,
                       do_sample=None):
    """Sampler::forward's random path (sampler.cpp:100-137) as TWO launches and no [B, V] temporary: apply_top_k_top_p in place
    on the logits in THEIR OWN dtype (what the reference does: the sampler sees the lm_head's 16-bit logits), then
    softmax_random_sample. Returns int32 [B]; `logits` holds the processed logits afterwards, as in the reference."""
    apply_top_k_top_p(logits, temperatures, top_k, top_p)
    return softmax_random_sample(logits, uniform, seed, offset, do_sample)


def _logits_2d(logits):
    _need_cuda(logits)
    if logits.dim() != 2 or logits.stride(1) != 1 or logits.dtype not in _DT:
        raise Mi355Error("logits: [batch, vocab] float32 / bfloat16 / float16 with unit column stride")
    return logits.size(0), logits.size(1), logits.stride(0)


def apply_penalties(logits, unique_token_ids, unique_token_counts, frequency_penalties=None, presence_penalties=None,
                    repetition_penalties=None):
    """apply_frequency_presence_penalties + apply_repetition_penalties (framework/sampling/logits_utils.cpp:24-52), in place on
    logits [B, V]; the order and the cast points of Sampler::forward (sampler.cpp:35-48)"""
    B, V, stride = _logits_2d(logits)
    if frequency_penalties is None and repetition_penalties is None:
        return logits
    U = unique_token_ids.size(1)
    ws = torch.empty(max(B * U, 1), dtype=torch.float32, device=logits.device)
    ids = unique_token_ids.to(torch.int64).contiguous()
    cnt = unique_token_counts.to(torch.int32).contiguous() if unique_token_counts is not None else None
    f32c = lambda t: None if t is None else t.to(torch.float32).contiguous()
    check(_lib.lib().xllm_mi355_apply_penalties(_p(logits), B, V, stride, _dt(logits), _p(ids), _p(cnt), U, _p(f32c(frequency_penalties)),
                                                _p(f32c(presence_penalties)), _p(f32c(repetition_penalties)), _p(ws), ws.numel() * 4,
                                                _stream()), "apply_penalties")
    return logits


def apply_temperatures(logits, temperatures):
    """apply_temperatures (logits_utils.cpp:54-64), in place"""
    B, V, stride = _logits_2d(logits)
    check(_lib.lib().xllm_mi355_apply_temperatures(_p(logits), B, V, stride, _dt(logits), _p(temperatures.to(torch.float32).contiguous()),
                                                   _stream()), "apply_temperatures")
    return logits


def apply_top_k_top_p(logits, temperatures=None, top_k=None, top_p=None):
    """apply_top_k_top_p (logits_utils.cpp:92-155), in place: temperatures, then top-k and / or top-p masking with -inf
    (sort-free: radix selection over keys and over fixed-point probability mass, xllm_mi355_apply_top_k_top_p)"""
    B, V, stride = _logits_2d(logits)
    f32c = lambda t: None if t is None else t.to(torch.float32).contiguous()
    k = None if top_k is None else top_k.to(torch.int64).contiguous()
    check(_lib.lib().xllm_mi355_apply_top_k_top_p(_p(logits), B, V, stride, _dt(logits), _p(f32c(temperatures)), _p(k), _p(f32c(top_p)),
                                                  _stream()), "apply_top_k_top_p")
    return logits


def greedy_argmax(logits: torch.Tensor) -> torch.Tensor:
    """Sampler::greedy_sample (framework/sampling/sampler.cpp:160-168): argmax over the last dim of [B, V] logits -> int64 [B];
    first index of the maximum, NaN above everything (torch.argmax)"""
    _need_cuda(logits)
    if logits.dim() != 2:
        raise Mi355Error("greedy_argmax: [batch, vocab] logits")
    if not logits.is_contiguous():
        logits = logits.contiguous()
    out = torch.empty(logits.size(0), dtype=torch.int64, device=logits.device)
    check(_lib.lib().xllm_mi355_greedy_argmax(_p(logits), _p(out), logits.size(0), logits.size(1), _dt(logits), _stream()),
          "greedy_argmax")
    return out


def rejection_sample(draft_token_ids, num_draft_tokens, cu_num_draft_tokens, draft_probs, target_probs,
                     bonus_token_ids, uniform_rand, uniform_probs):
    """dcu::rejection_sample (kernels/dcu/rejection_sample.hip:141-215) -> int32 [batch + total_drafts]"""
    _need_cuda(draft_token_ids, num_draft_tokens, cu_num_draft_tokens, draft_probs, target_probs, bonus_token_ids,
               uniform_rand, uniform_probs)
    f = lambda t: t.to(torch.float32).contiguous()
    dp, tp, ur, up = f(draft_probs), f(target_probs), f(uniform_rand), f(uniform_probs)
    if dp.shape != tp.shape or up.shape != tp.shape or dp.dim() != 2:
        raise Mi355Error("draft_probs / target_probs / uniform_probs must be [total_drafts, vocab]")
    B = num_draft_tokens.numel()
    out = torch.empty(B + draft_token_ids.numel(), dtype=torch.int32, device=tp.device)
    draft_token_ids_c = draft_token_ids.contiguous()  # keep the (possibly new) tensor alive across the call
    check(_lib.lib().xllm_mi355_rejection_sample(
        _p(draft_token_ids_c), _p(num_draft_tokens), _p(cu_num_draft_tokens), _p(dp), _p(tp),
        _p(bonus_token_ids), _p(ur), _p(up), B, tp.size(1), _p(out), _stream()), "rejection_sample")
    return out


def moe_fused_topk(gating_output, topk: int, renormalize: bool, correction_bias=None, scoring_func: str = "softmax"):
    """cuda::moe_fused_topk (kernels/cuda/moe/moe_fused_topk.cu:31-61) -> (topk_weights f32, topk_ids int32)"""
    _need_cuda(gating_output)
    if scoring_func not in ("softmax", "sigmoid"):
        raise Mi355Error(f"Unsupported scoring function for moe topk: {scoring_func}")   # reference LOG(FATAL) :55
    T, E = gating_output.shape
    g = gating_output.contiguous()
    w = torch.empty(T, topk, dtype=torch.float32, device=g.device)
    ids = torch.empty(T, topk, dtype=torch.int32, device=g.device)
    bias = correction_bias.to(torch.float32).contiguous() if (correction_bias is not None and scoring_func == "sigmoid") else None
    check(_lib.lib().xllm_mi355_moe_fused_topk(_p(g), _dt(g), T, E, topk, int(renormalize), _p(bias),
                                               0 if scoring_func == "softmax" else 1, _p(w), _p(ids), _stream()),
          "moe_fused_topk")
    return w, ids


def moe_grouped_topk(gating_output, topk: int, num_expert_group: int, topk_group: int, renormalize: bool,
                     correction_bias=None, scoring_func: str = "softmax", routed_scaling_factor: float = 1.0):
    """dcu::moe_grouped_topk (kernels/dcu/topk_gate.cpp:59-125) -> (topk_weights f32, topk_ids int32)"""
    _need_cuda(gating_output)
    if scoring_func not in ("softmax", "sigmoid"):
        raise Mi355Error(f"moe_grouped_topk: unsupported scoring function {scoring_func}")          # :81-82
    if correction_bias is not None and scoring_func != "sigmoid":
        raise Mi355Error("moe_grouped_topk: correction bias is supported only for sigmoid scoring")  # :96-98
    T, E = gating_output.shape
    g = gating_output.contiguous()
    w = torch.empty(T, topk, dtype=torch.float32, device=g.device)
    ids = torch.empty(T, topk, dtype=torch.int32, device=g.device)
    bias = correction_bias.to(torch.float32).contiguous() if correction_bias is not None else None
    check(_lib.lib().xllm_mi355_moe_grouped_topk(_p(g), _dt(g), T, E, topk, num_expert_group, topk_group, int(renormalize),
                                                 _p(bias), 0 if scoring_func == "softmax" else 1,
                                                 float(routed_scaling_factor), _p(w), _p(ids), _stream()),
          "moe_grouped_topk")
    return w, ids


def moe_active_topk(gating_output, topk: int, num_expert_group: int, topk_group: int, renormalize: bool,
                    correction_bias=None, scoring_func: str = "softmax", routed_scaling_factor: float = 1.0):
    """dcu::moe_active_topk (kernels/dcu/topk_gate.cpp:127-146): grouped gate when num_expert_group > 1, else the plain
    fused top-k (which, as in the reference, takes no routed_scaling_factor)"""
    if num_expert_group > 1:
        return moe_grouped_topk(gating_output, topk, num_expert_group, topk_group, renormalize, correction_bias,
                                scoring_func, routed_scaling_factor)
    return moe_fused_topk(gating_output, topk, renormalize, correction_bias, scoring_func)
