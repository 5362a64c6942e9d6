"""ctypes binding of include/xllm_mi355.h.  There is NO fallback: if the HIP library is missing or a
call returns non-zero this raises (the reference aborts with CHECK/TORCH_CHECK in the same places)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XLLM_MI355_LIB selects an alternative build of the same ABI (A/B experiments); default is the in-tree library
LIB_PATH = os.environ.get("XLLM_MI355_LIB") or os.path.join(_HERE, "lib", "libxllm_mi355.so")

vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
ci = C.c_int
u64 = C.c_uint64



class DecodeMetadata(C.Structure):
    """xllm_mi355_decode_metadata_t (include/xllm_mi355.h)"""
    _fields_ = ([(n, vp) for n in (
        "src_tokens", "src_positions", "src_new_cache_slots", "src_kv_seq_lens", "src_paged_kv_indptr",
        "src_paged_kv_indices", "src_paged_kv_last_page_len", "dst_tokens", "dst_positions", "dst_new_cache_slots",
        "dst_kv_seq_lens", "dst_kv_seq_lens_delta", "dst_paged_kv_indptr", "dst_paged_kv_indices",
        "dst_paged_kv_last_page_len")] +
        [(n, i64) for n in ("actual_num_tokens", "padded_num_tokens", "actual_batch_size", "actual_indices_size")] +
        [("dst_block_table", vp), ("dst_kv_lens", vp), ("max_blocks_per_seq", i64), ("padded_batch_size", i64)])



class HostBufferEntry(C.Structure):
    """xllm_mi355_host_buffer_entry_t (include/xllm_mi355.h)"""
    _fields_ = [("data", vp), ("bytes", u64), ("offset", u64), ("aligned_bytes", u64)]


class HostBatch(C.Structure):
    """xllm_mi355_host_batch_t (include/xllm_mi355.h)"""
    _fields_ = ([(n, i64) for n in ("cap_tokens", "cap_indices", "cap_sequences", "cap_block_table")] +
                [(n, vp) for n in ("new_cache_slots", "positions", "paged_kv_indptr", "paged_kv_indices",
                                   "paged_kv_last_page_len", "block_tables", "q_cu_seq_lens", "kv_cu_seq_lens",
                                   "q_seq_lens", "kv_seq_lens")] +
                [(n, i32) for n in ("num_sequences", "q_max_seq_len", "kv_max_seq_len")] +
                [(n, i64) for n in ("n_tokens", "n_indices", "max_blocks", "total_kv_len")])


_SIGS = {
    "xllm_mi355_host_cache_slots": ([vp, i64, i64, i64, i64, vp], ci),
    "xllm_mi355_host_plan_input_buffer": ([C.POINTER(HostBufferEntry), i64, u64, C.POINTER(u64)], ci),
    "xllm_mi355_host_pack_input_buffer": ([C.POINTER(HostBufferEntry), i64, vp, u64], ci),
    "xllm_mi355_host_build_batch": ([vp, vp, vp, vp, i64, i64, C.POINTER(HostBatch)], ci),
    "xllm_mi355_abi_version": ([], ci),
    "xllm_mi355_build_digest": ([], C.c_char_p),
    "xllm_mi355_block_copy": ([vp, vp, vp, vp, vp, i64, i64, i64, i64, vp], ci),
    "xllm_mi355_bmm_heads": ([vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_gemm_plan_hint": ([ci, ci, ci], None),
    "xllm_mi355_scaled_matmul_rope_cache_packed": ([vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp, vp, vp, vp, vp, i64, i64, i64,
                                                    i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_oneshot_allreduce_buffer_bytes": ([sz], sz),
    "xllm_mi355_ipc_alloc": ([sz, C.POINTER(vp), C.POINTER(ci)], ci),
    "xllm_mi355_ipc_free": ([vp], ci),
    "xllm_mi355_ipc_get_handle": ([vp, vp], ci),
    "xllm_mi355_ipc_open_handle": ([vp, C.POINTER(vp)], ci),
    "xllm_mi355_ipc_close_handle": ([vp], ci),
    "xllm_mi355_oneshot_allreduce": ([vp, i64, ci, C.POINTER(vp), ci, ci, sz, vp, vp, C.c_double, vp], ci),
    "xllm_mi355_oneshot_allreduce_add_rms_norm": ([vp, vp, vp, f32, vp, vp, vp, vp, i64, i64, ci, C.POINTER(vp), ci, ci, sz, vp, vp,
                                                   C.c_double, ci, vp], ci),
    "xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm": (
        [vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, C.POINTER(vp), ci, ci, sz, vp, vp,
         C.c_double, ci, vp], ci),
    "xllm_mi355_decode_metadata_update": ([C.POINTER(DecodeMetadata), vp], ci),
    "xllm_mi355_reshape_paged_cache": ([vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_build_block_table_from_paged_kv": ([vp, vp, i32, i32, vp, vp], ci),
    "xllm_mi355_rms_norm": ([vp, vp, vp, f32, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_fused_add_rms_norm": ([vp, vp, vp, f32, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_rms_norm_static_fp8_quant": ([vp, vp, vp, vp, vp, f32, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_rms_norm_dynamic_int8_quant": ([vp, vp, vp, vp, vp, f32, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_rotary_embedding": ([vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, ci, ci, vp], ci),
    "xllm_mi355_fused_qk_norm_rope": ([vp, i64, i64, i64, i64, i64, f32, vp, vp, vp, ci, ci, vp, ci, vp], ci),
    "xllm_mi355_act_and_mul": ([vp, vp, i64, i64, ci, ci, vp], ci),
    "xllm_mi355_act_and_mul_dynamic_int8_quant": ([vp, vp, vp, i64, i64, ci, ci, vp], ci),
    "xllm_mi355_act_and_mul_dynamic_int8_quant_live": ([vp, vp, vp, i64, i64, ci, ci, vp, i64, vp], ci),
    "xllm_mi355_scaled_quantize": ([vp, vp, vp, i64, i64, ci, vp], ci),
    "xllm_mi355_scaled_matmul": ([vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_scaled_matmul_add": ([vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_add16": ([vp, vp, vp, i64, ci, vp], ci),
    "xllm_mi355_set_gemm_workspace": ([vp, sz], ci),
    "xllm_mi355_set_gemm_workspace_for_stream": ([vp, vp, sz], ci),
    "xllm_mi355_scaled_matmul_add_rms_norm": ([vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_pack_weight_i8": ([vp, vp, i64, i64, vp], ci),
    "xllm_mi355_scaled_matmul_packed": ([vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_scaled_matmul_add_rms_norm_packed": ([vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i64, i64, i64, ci, vp, sz,
                                                      vp], ci),
    "xllm_mi355_scaled_matmul_gate_up_act": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_quantize_with_row_amax": ([vp, vp, vp, vp, i64, i64, ci, vp], ci),
    "xllm_mi355_static_scaled_fp8_quant": ([vp, vp, vp, i64, ci, vp], ci),
    "xllm_mi355_fp8_scaled_quantize": ([vp, vp, vp, vp, i64, ci, vp], ci),
    "xllm_mi355_fp8_scaled_quantize_ws": ([vp, vp, vp, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_fp8_scaled_matmul": ([vp, vp, vp, i64, vp, i64, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_pack_weight_fp8": ([vp, vp, i64, i64, vp], ci),
    "xllm_mi355_fp8_scaled_matmul_packed": ([vp, vp, vp, i64, vp, i64, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_matmul": ([vp, vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_pack_weight_16": ([vp, vp, i64, i64, vp], ci),
    "xllm_mi355_matmul_packed": ([vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_matmul_gate_up_act": ([vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_matmul_argmax_workspace_bytes": ([i64, i64], sz),
    "xllm_mi355_matmul_argmax_packed": ([vp, vp, vp, vp, vp, i64, i64, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_prefill_attention": ([vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, ci, i64,
                                      ci, vp], ci),
    "xllm_mi355_paged_decode_attention_int8": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64,
                                                f32, i64, ci, vp], ci),
    "xllm_mi355_paged_decode_attention_int8_ws": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64,
                                                f32, i64, ci, vp, sz, vp], ci),
    "xllm_mi355_rotary_embedding_and_cache": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64,
                                               i64, i64, ci, ci, vp], ci),
    "xllm_mi355_paged_attention_workspace_bytes": ([i64, i64, i64, i64, i64], sz),
    "xllm_mi355_paged_attention": ([vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64,
                                    i64, f32, ci, i64, ci, vp, sz, vp], ci),
}
# entry points declared in the header but implemented in a later build step are bound lazily
_OPTIONAL = {
    "xllm_mi355_mla_decode": ([vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, ci, vp, sz, vp], ci),
    "xllm_mi355_mla_prefill": ([vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, f32, ci, ci, vp, sz,
                                vp], ci),
    "xllm_mi355_random_sample": ([vp, vp, i64, i64, vp, u64, u64, vp], ci),
    "xllm_mi355_philox_uniform": ([vp, i64, u64, u64, vp], ci),
    "xllm_mi355_softmax_random_sample": ([vp, vp, i64, i64, i64, ci, vp, u64, u64, vp, vp], ci),
    "xllm_mi355_rejection_sample": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, vp], ci),
    "xllm_mi355_greedy_argmax": ([vp, vp, i64, i64, ci, vp], ci),
    "xllm_mi355_apply_penalties_workspace_bytes": ([i64, i64], sz),
    "xllm_mi355_apply_penalties": ([vp, i64, i64, i64, ci, vp, vp, i64, vp, vp, vp, vp, sz, vp], ci),
    "xllm_mi355_apply_temperatures": ([vp, i64, i64, i64, ci, vp, vp], ci),
    "xllm_mi355_apply_top_k_top_p": ([vp, i64, i64, i64, ci, vp, vp, vp, vp], ci),
    "xllm_mi355_moe_fused_topk": ([vp, ci, i64, i64, i64, ci, vp, ci, vp, vp, vp], ci),
    "xllm_mi355_moe_grouped_topk": ([vp, ci, i64, i64, i64, i64, i64, ci, vp, ci, f32, vp, vp, vp], ci),
    "xllm_mi355_set_moe_workspace": ([vp, sz], ci),
    "xllm_mi355_set_moe_workspace_for_stream": ([vp, vp, sz], ci),
    "xllm_mi355_moe_compute_index": ([vp, i64, i64, i64, vp, vp, vp, vp], ci),
    "xllm_mi355_moe_combine": ([vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_moe_combine_sorted": ([vp, vp, vp, vp, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_moe_combine_sorted_local": ([vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_group_gemm": ([vp, vp, vp, vp, i64, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_group_gemm_w8a8": ([vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, ci, vp], ci),
    "xllm_mi355_group_gemm_gather": ([vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, i64, ci, vp], ci),
}

_lib = None
# the ABI version these bindings were written for (XLLM_MI355_ABI_VERSION of include/xllm_mi355.h): a library that reports
# another one has different argument lists behind the same names (version 1 -> 2 inserted `grid_limit` before `stream` in two
# entry points), so it is refused at load time rather than called
ABI_VERSION = 2


class Mi355Error(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        l.xllm_mi355_abi_version.argtypes, l.xllm_mi355_abi_version.restype = [], ci
        got = l.xllm_mi355_abi_version()
        if got != ABI_VERSION:
            raise Mi355Error(f"{LIB_PATH} reports ABI version {got}, these bindings are written for {ABI_VERSION}: "
                             "rebuild it (`make -C xllm_amd/csrc`)")
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes, fn.restype = args, res
        for name, (args, res) in _OPTIONAL.items():
            if hasattr(l, name):
                fn = getattr(l, name)
                fn.argtypes, fn.restype = args, res
        l.xllm_mi355_strerror.argtypes, l.xllm_mi355_strerror.restype = [ci], C.c_char_p
        _lib = l
    return _lib


def exported_symbols():
    return list(_SIGS) + list(_OPTIONAL) + ["xllm_mi355_strerror"]


def check(rc: int, what: str):
    if rc != 0:
        raise Mi355Error(f"{what}: {lib().xllm_mi355_strerror(rc).decode()} (rc={rc})")
