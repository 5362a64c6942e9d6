"""The callers of the hot path, in the reference's order, over the xllm_amd operators.

Reference: Qwen2DecoderLayerImpl::forward (xllm/core/layers/qwen2_decoder_layer.cpp:87-110),
Qwen2AttentionImpl::forward (layers/common/qwen2_attention.cpp:132-193), DenseMLPImpl::forward
(layers/common/dense_mlp.cpp:97-116), the w8a8-dynamic linear (layers/common/linear.cpp:481-507: scaled_quantize
then scaled_matmul), Row/Column-parallel sharding (linear.cpp:616-716, 1405-1522) and
LlmModelImplBase::forward (models/llm/llm_model_base.h:60-125) for the layer loop + final norm + lm_head;
FusedMoEImpl::forward_experts (layers/dcu/fused_moe.cpp:143-337) for the routed-expert FFN;
DeepseekV2AttentionImpl::forward (layers/dcu/deepseek_v2_attention.cpp:264-317) for MLA.

Weights are synthetic (random-init of the architecture): this module is the fixed-shape harness of SURVEY 8d,
not a checkpoint loader.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops, parallel
from .attention import AttentionImpl, AttentionMetadata, KVCache


@dataclass
class ModelArgs:
    hidden_size: int = 3584
    n_layers: int = 28
    n_heads: int = 28
    n_kv_heads: int = 4
    head_dim: int = 128
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 32768

    @staticmethod
    def qwen2_7b():  # xllm/models/llm/qwen2.h:89-121
        return ModelArgs()

    @staticmethod
    def qwen2_0_5b():
        return ModelArgs(896, 24, 14, 2, 64, 4864, 151936, 1e-6, 1e6, 32768)


def build_cos_sin_cache(args: ModelArgs, dtype, device, max_pos: Optional[int] = None):
    """layers/common/rotary_embedding_util.cpp:157-192 + rotary_embedding.cpp:46-52: [cos(rot/2) || sin(rot/2)]"""
    rot = args.head_dim
    n = max_pos or args.max_position_embeddings
    inv_freq = 1.0 / torch.pow(torch.tensor(args.rope_theta, dtype=torch.float32),
                               torch.arange(0, rot, 2, dtype=torch.float32) / rot)
    fr = torch.outer(torch.arange(n, dtype=torch.float32), inv_freq)
    return torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(device)


def _shard(t: torch.Tensor, dim: int, rank: int, world: int) -> torch.Tensor:
    n = t.size(dim) // world
    return t.narrow(dim, rank * n, n).contiguous()


class QuantLinear:
    """Column/Row-parallel linear, w8a8-dynamic (int8) / fp8 / 16-bit, weight [N_local, K_local].

    Parameters are drawn for the FULL layer from `gen` (every rank of a TP group passes a generator in the same state, so
    every rank draws the same tensors) and then cut: `shard=("col", rank, world)` keeps output rows (ColumnParallelLinear,
    linear.cpp:616-716; `col_blocks` = the fused projections [q; k; v] / [gate; up] whose blocks are sharded one by one,
    with `kv_replicas` for kv heads shared by several ranks, qwen2_attention.cpp:57-65), `shard=("row", rank, world)` keeps
    input columns (RowParallelLinear, linear.cpp:1405-1522: bias only on rank 0 so that the all-reduce adds it once).
    int8 scales are per output channel over the FULL K (a checkpoint is quantised before it is sharded)."""

    def __init__(self, n: int, k: int, bias: bool, mode: str, dtype, device, gen, row_parallel_pg=None, shard=None,
                 col_blocks=None, pack16: bool = True):
        self.mode, self.dtype, self.pg = mode, dtype, row_parallel_pg
        self.weight_packed = None
        if mode == "int8":
            # random-init weights of the architecture (N(0, initializer_range = 0.02), models/llm/qwen2.h) put through
            # the symmetric per-output-channel int8 quantisation a W8A8 checkpoint carries: w_q = round(w / s),
            # s = amax_row / 127. (A uniform draw over [-127, 127] is not what any quantised checkpoint looks like.)
            w = torch.empty(n, k, device=device, dtype=torch.float32).normal_(0.0, 0.02, generator=gen)
            self.w_scale = (w.abs().amax(dim=1) / 127.0).clamp_min(1e-12)
            self.weight = torch.round(w / self.w_scale[:, None]).clamp_(-127, 127).to(torch.int8)
            del w
        elif mode == "fp8":
            self.weight = (torch.randn(n, k, device=device, generator=gen)).to(torch.float8_e4m3fn)
            self.w_scale = torch.full((1,), 1.0 / math.sqrt(k), device=device)
        else:
            self.weight = (torch.randn(n, k, device=device, generator=gen) / math.sqrt(k)).to(dtype)
        self.bias = (torch.randn(n, device=device, generator=gen) * 0.1).to(dtype) if bias else None
        if shard is not None and shard[2] > 1:
            kind, rank, world = shard
            if kind == "col":
                blocks = col_blocks or [(n, world, rank)]          # (rows of the block, ways it is cut, this rank's piece)
                rows, off = [], 0
                for size, ways, piece in blocks:
                    step = size // ways
                    rows.append(torch.arange(off + piece * step, off + (piece + 1) * step, device=device))
                    off += size
                idx = torch.cat(rows)
                if self.weight.element_size() == 1 and self.weight.dtype != torch.int8:   # float8: index_select is not implemented
                    self.weight = self.weight.view(torch.uint8).index_select(0, idx).contiguous().view(self.weight.dtype)
                else:
                    self.weight = self.weight.index_select(0, idx).contiguous()
                if self.bias is not None:
                    self.bias = self.bias.index_select(0, idx).contiguous()
                if mode == "int8":
                    self.w_scale = self.w_scale.index_select(0, idx).contiguous()
            else:
                if self.weight.element_size() == 1 and self.weight.dtype != torch.int8:
                    self.weight = _shard(self.weight.view(torch.uint8), 1, rank, world).view(self.weight.dtype)
                else:
                    self.weight = _shard(self.weight, 1, rank, world)
                if self.bias is not None and rank != 0:
                    self.bias = torch.zeros_like(self.bias)
        if mode == "int8" and self.weight.is_cuda:
            # decode-shaped GEMMs stream the weights in MFMA-fragment order (xllm_mi355_pack_weight_i8, once at load time);
            # the row-major copy stays for the prefill kernels
            self.weight_packed = ops.pack_weight_i8(self.weight)
        elif mode == "fp8" and self.weight.is_cuda:
            self.weight_packed = ops.pack_weight_fp8(self.weight)
        elif self.weight.is_cuda and self.weight.dtype in (torch.bfloat16, torch.float16) and pack16:
            self.weight_packed = ops.pack_weight_16(self.weight)     # (cfg2: BASELINE config 2 runs 16-bit linears at M = 64)

    def forward(self, x, pre_quant=None, reduce: bool = True):
        """`reduce=False`: a row-parallel shard returns its PARTIAL sums (the caller fuses the all-reduce with what follows)"""
        if self.mode == "int8":
            q, s = pre_quant if pre_quant is not None else ops.scaled_quantize(x)
            y = ops.scaled_matmul(q, self.weight, s, self.w_scale, self.dtype, self.bias, b_packed=self.weight_packed)
        elif self.mode == "fp8":
            q, s = pre_quant if pre_quant is not None else ops.fp8_scaled_quantize(x)
            y = ops.fp8_scaled_matmul(q, self.weight, s, self.w_scale, self.dtype, self.bias, b_packed=self.weight_packed)
        else:
            y = ops.matmul(x, self.weight, self.bias, b_packed=self.weight_packed)
        return parallel.reduce(y, self.pg) if (self.pg is not None and reduce) else y

    def weight_bytes(self) -> int:
        return self.weight.numel() * self.weight.element_size()


class Qwen2DecoderLayer:
    def __init__(self, args: ModelArgs, mode: str, dtype, device, gen, tp: Optional[parallel.ProcessGroup] = None,
                 fuse: bool = True):
        tp_size = tp.world_size() if tp else 1
        tp_rank = tp.rank() if tp else 0
        assert args.n_heads % tp_size == 0  # qwen2_attention.cpp:54
        self.nq = args.n_heads // tp_size
        if args.n_kv_heads >= tp_size:       # qwen2_attention.cpp:57-65
            assert args.n_kv_heads % tp_size == 0
            self.nkv, kv_ways, kv_piece = args.n_kv_heads // tp_size, tp_size, tp_rank
        else:                                # kv heads replicated: rank r holds kv head r / (tp / n_kv_heads)
            assert tp_size % args.n_kv_heads == 0
            self.nkv, kv_ways, kv_piece = 1, args.n_kv_heads, tp_rank // (tp_size // args.n_kv_heads)
        self.d, self.args, self.mode, self.fuse, self.dtype = args.head_dim, args, mode, fuse and mode == "int8", dtype
        self.q_size, self.kv_size = self.nq * self.d, self.nkv * self.d
        H, I_full = args.hidden_size, args.intermediate_size
        I = I_full // tp_size
        self.I = I
        # replicated parameters and the full tensors of the sharded ones: the same draws on every rank of the TP group
        w = lambda: (torch.rand(H, device=device, generator=gen) + 0.5).to(dtype)
        self.input_norm_w, self.post_norm_w = w(), w()
        qf, kvf = args.n_heads * self.d, args.n_kv_heads * self.d
        col, row = ("col", tp_rank, tp_size), ("row", tp_rank, tp_size)
        self.qkv_proj = QuantLinear(qf + 2 * kvf, H, True, mode, dtype, device, gen, shard=col,
                                    col_blocks=[(qf, tp_size, tp_rank), (kvf, kv_ways, kv_piece), (kvf, kv_ways, kv_piece)])
        self.o_proj = QuantLinear(H, qf, False, mode, dtype, device, gen, row_parallel_pg=tp, shard=row)
        self.gate_up_proj = QuantLinear(2 * I_full, H, False, mode, dtype, device, gen, shard=col,
                                        col_blocks=[(I_full, tp_size, tp_rank), (I_full, tp_size, tp_rank)])
        self.down_proj = QuantLinear(H, I_full, False, mode, dtype, device, gen, row_parallel_pg=tp, shard=row)
        self.attn = AttentionImpl(self.nq, self.d, math.sqrt(1.0 / self.d), self.nkv)

    def weight_bytes(self) -> int:
        return sum(l.weight_bytes() for l in (self.qkv_proj, self.o_proj, self.gate_up_proj, self.down_proj))

    def _norm(self, x, residual, w):
        """apply_norm (qwen2_decoder_layer.cpp:66-85); returns (normed or pre-quantised, residual)"""
        eps = self.args.rms_norm_eps
        if self.fuse:  # N1: norm (+add) + per-token int8 quant in one pass
            if residual is None:
                return ops.rms_norm_dynamic_int8_quant(x, w, eps), x
            return ops.rms_norm_dynamic_int8_quant(x, w, eps, residual=residual), residual
        if residual is None:
            out = torch.empty_like(x)
            ops.rms_norm(out, x, w, eps)
            return out, x
        ops.fused_add_rms_norm(x, residual, w, eps)
        return x, residual

    def _fused_linear_norm(self, lin: "QuantLinear", pre_quant, residual, norm_w, quantize=True):
        """N1 across the GEMM boundary: row-parallel W8A8 linear + residual add + RMSNorm (+ int8 quant) in two launches
        (ops.scaled_matmul_add_rms_norm). Only without a TP all-reduce in between; None = not applicable."""
        # (a group whose exchange is stubbed -- bench.py --emulate-tp: one rank's compute, no peers -- takes this path too: it is
        # the one-shot kernel of _tp_linear_norm minus the peer reads, GEMM -> ONE consumer of the int32 slabs)
        exchanged = lin.pg is not None and lin.pg.world_size() > 1 and not getattr(lin.pg, "exchange_stubbed", False)
        if not self.fuse or exchanged or lin.mode != "int8" or residual is None or norm_w is None:
            return None
        return ops.scaled_matmul_add_rms_norm(pre_quant[0], lin.weight, pre_quant[1], lin.w_scale, residual, norm_w,
                                              self.args.rms_norm_eps, lin.bias, quantize=quantize, b_packed=lin.weight_packed)

    def _gate_up_act_quant(self, h):
        """gate_up_proj + SiLU * mul + the int8 quant of down_proj's operand (dense_mlp.cpp:97-116): the GEMM epilogue computes the
        activation (ops.scaled_matmul_silu_mul_quant, round 3); the unfused pair of operators when the shape is outside its envelope"""
        lin = self.gate_up_proj
        if lin.mode == "int8" and lin.bias is None:
            out = ops.scaled_matmul_silu_mul_quant(h[0], lin.weight, h[1], lin.w_scale, self.dtype, None, b_packed=lin.weight_packed)
            if out is not None:
                return out
        return ops.act_and_mul_dynamic_int8_quant(lin.forward(None, pre_quant=h), "silu")

    def _tp_linear_norm(self, lin: "QuantLinear", pre_quant, residual, norm_w, quantize=True):
        """Tensor parallel (round 3): row-parallel W8A8 linear -> ONE kernel for the SUM all-reduce over xGMI + residual add +
        RMSNorm (+ int8 quant) (ProcessGroup.allreduce_add_rms_norm = xllm_mi355_oneshot_allreduce_add_rms_norm), so a TP
        half-layer is GEMM -> one kernel instead of GEMM -> collective -> row-wise kernel (linear.cpp:1518-1520 +
        qwen2_decoder_layer.cpp:66-85). None = not applicable (no TP group, one-shot path off, message too large)."""
        if not self.fuse or lin.pg is None or lin.pg.world_size() == 1 or lin.pg.oneshot is None or residual is None \
                or norm_w is None:
            return None
        if pre_quant[0].size(0) * lin.weight.size(0) * 2 > lin.pg.oneshot.max_bytes:
            return None       # (a prefill-sized message: decided before the GEMM runs, the caller takes linear + reduce)
        if lin.mode == "int8" and lin.weight_packed is not None and pre_quant[0].dim() == 2:
            # the GEMM's int32 K-slice sums go straight into the one-shot kernel (no dequant pass, round 3)
            out = lin.pg.matmul_allreduce_add_rms_norm(pre_quant[0], pre_quant[1], lin.weight_packed, lin.w_scale, lin.bias,
                                                       residual, norm_w, self.args.rms_norm_eps, quantize)
            if out is not None:
                return out
        y = lin.forward(None, pre_quant=pre_quant, reduce=False)      # this rank's partial sums, 16 bit (bias on rank 0 only)
        return lin.pg.allreduce_add_rms_norm(y, residual, norm_w, self.args.rms_norm_eps, quantize) if y.dim() == 2 else None

    # ---- the decode step cut at the attention kernel (dual micro-batch executor, DualBatchDecoder below) ----------
    def _qkv_rope_cache(self, h, positions, md: AttentionMetadata, kv_cache: KVCache, cos_sin):
        """N1 across the GEMM boundary (small M, packed weights): W8A8 qkv projection -> dequant -> RoPE -> KV write in two
        launches (ops.scaled_matmul_rope_cache), bit-identical to qkv_proj + rotary_embedding_and_cache; None = not applicable"""
        if not self.fuse or self.qkv_proj.weight_packed is None or positions.dtype != torch.int64:
            return None
        return ops.scaled_matmul_rope_cache(h[0], self.qkv_proj.weight_packed, h[1], self.qkv_proj.w_scale, self.qkv_proj.bias,
                                            positions, cos_sin, md.slot_mapping, kv_cache.k_cache, kv_cache.v_cache, self.nq,
                                            self.nkv, self.d, self.dtype)

    def pre_attention(self, x, residual, positions, md: AttentionMetadata, kv_cache: KVCache, cos_sin, h_in=None):
        """input norm (unless the previous layer fused it) + qkv_proj + RoPE + KV write; returns (q, residual)"""
        if h_in is not None:
            h = h_in
        else:
            h, residual = self._norm(x, residual, self.input_norm_w)
        qkv = self._qkv_rope_cache(h, positions, md, kv_cache, cos_sin)
        if qkv is not None:
            return qkv[:, :self.q_size], residual
        qkv = self.qkv_proj.forward(None, pre_quant=h) if self.fuse else self.qkv_proj.forward(h)
        q = qkv[:, :self.q_size]
        k = qkv[:, self.q_size:self.q_size + self.kv_size]
        v = qkv[:, self.q_size + self.kv_size:]
        ops.rotary_embedding_and_cache(positions, q, k, v, cos_sin, md.slot_mapping, kv_cache.k_cache, kv_cache.v_cache,
                                       self.d, True)
        return q, residual

    def attention_kernel(self, q, md: AttentionMetadata, kv_cache: KVCache):
        """the HBM-bound kernel alone; returns the int8 operand of o_proj (q8, scale)"""
        fused = ops.paged_decode_attention_int8(q.unflatten(-1, (self.nq, self.d)), kv_cache.k_cache, kv_cache.v_cache,
                                                md.kv_seq_lens, md.block_table, md.max_seq_len, self.attn.scale,
                                                self.attn.window_left)
        if fused is not None:
            return fused[0], fused[1]      # (the third element is the 16-bit output; QuantLinear.forward unpacks a pair)
        attn = ops.paged_attention(q.unflatten(-1, (self.nq, self.d)), kv_cache.k_cache, kv_cache.v_cache, None,
                                   md.kv_seq_lens, md.block_table, 1, md.max_seq_len, self.attn.scale, False,
                                   self.attn.window_left)
        return ops.scaled_quantize(attn)

    def post_attention(self, o_in, residual, next_norm_w, next_quant):
        """o_proj (+ post norm) + MLP (+ the next layer's input norm); returns (x, residual, h_next) like forward()"""
        h = self._fused_linear_norm(self.o_proj, o_in, residual, self.post_norm_w)
        if h is None:
            h = self._tp_linear_norm(self.o_proj, o_in, residual, self.post_norm_w)
        if h is None:
            x = self.o_proj.forward(None, pre_quant=o_in)
            h, residual = self._norm(x, residual, self.post_norm_w)
        act_q = self._gate_up_act_quant(h)
        h_next = self._fused_linear_norm(self.down_proj, act_q, residual, next_norm_w, quantize=next_quant)
        if h_next is None:
            h_next = self._tp_linear_norm(self.down_proj, act_q, residual, next_norm_w, quantize=next_quant)
        if h_next is not None:
            return None, residual, h_next
        return self.down_proj.forward(None, pre_quant=act_q), residual, None

    def forward(self, x, residual, positions, md: AttentionMetadata, kv_cache: KVCache, cos_sin, h_in=None,
                next_norm_w=None, next_quant=True):
        """returns (x, residual, h_next): h_next is the NEXT layer's (or the model's final) norm output when this
        layer's down_proj was fused with it (then x is None), else None."""
        if h_in is not None:
            h = h_in                      # the previous layer already ran this layer's input norm (fused)
        else:
            h, residual = self._norm(x, residual, self.input_norm_w)
        qkv = self._qkv_rope_cache(h, positions, md, kv_cache, cos_sin)   # small M: GEMM -> dequant + RoPE + KV write fused
        rope_done = qkv is not None
        if qkv is None:
            qkv = self.qkv_proj.forward(None, pre_quant=h) if self.fuse else self.qkv_proj.forward(h)
        q = qkv[:, :self.q_size]
        k = qkv[:, self.q_size:self.q_size + self.kv_size]
        v = qkv[:, self.q_size + self.kv_size:]
        decode = not (md.is_prefill or md.is_chunked_prefill)
        fused_attn = None
        if self.fuse and decode:
            # N1 fusions on the decode path: RoPE + KV write in one launch, and the attention epilogue emits the
            # int8 operand of o_proj directly (falls back when the batch is small enough to need split-KV)
            if not rope_done:
                ops.rotary_embedding_and_cache(positions, q, k, v, cos_sin, md.slot_mapping, kv_cache.k_cache,
                                               kv_cache.v_cache, self.d, True)
            fused_attn = ops.paged_decode_attention_int8(q.unflatten(-1, (self.nq, self.d)), kv_cache.k_cache,
                                                         kv_cache.v_cache, md.kv_seq_lens, md.block_table,
                                                         md.max_seq_len, self.attn.scale, self.attn.window_left)
            if fused_attn is None:
                attn = ops.paged_attention(q.unflatten(-1, (self.nq, self.d)), kv_cache.k_cache, kv_cache.v_cache, None,
                                           md.kv_seq_lens, md.block_table, 1, md.max_seq_len, self.attn.scale, False,
                                           self.attn.window_left)
        elif self.fuse:  # prefill / chunked prefill: the same RoPE + KV-write fusion, then the attention kernel alone
            if not rope_done:
                ops.rotary_embedding_and_cache(positions, q, k, v, cos_sin, md.slot_mapping, kv_cache.k_cache,
                                               kv_cache.v_cache, self.d, True)
            attn, _ = self.attn.forward(md, q, k, v, kv_cache, kv_written=True)
        else:
            ops.rotary_embedding(positions, q, k, cos_sin, True, head_size=self.d)
            attn, _ = self.attn.forward(md, q, k, v, kv_cache)
        h = None
        if self.fuse and decode:
            o_in = (fused_attn[0], fused_attn[1]) if fused_attn is not None else ops.scaled_quantize(attn)
            h = self._fused_linear_norm(self.o_proj, o_in, residual, self.post_norm_w)   # residual updated in place
            if h is None:   # tensor parallel: GEMM -> all-reduce + add + norm + quant in one kernel
                h = self._tp_linear_norm(self.o_proj, o_in, residual, self.post_norm_w)
            if h is None:
                x = self.o_proj.forward(None, pre_quant=o_in)
        elif fused_attn is not None:
            x = self.o_proj.forward(None, pre_quant=(fused_attn[0], fused_attn[1]))
        else:
            x = self.o_proj.forward(attn)
        if h is None:
            h, residual = self._norm(x, residual, self.post_norm_w)
        act16 = None if self.fuse else self._gate_up_act_16(h)   # unquantised layer, decode shapes: SiLU * mul in the GEMM epilogue
        if act16 is not None:
            return self.down_proj.forward(act16), residual, None
        gate_up = None if self.fuse else self.gate_up_proj.forward(h)
        if self.fuse:  # N1: gate_up GEMM with silu*mul in its epilogue + int8 quant feeding down_proj
            act_q = self._gate_up_act_quant(h)
            if decode:
                h_next = self._fused_linear_norm(self.down_proj, act_q, residual, next_norm_w, quantize=next_quant)
                if h_next is None:
                    h_next = self._tp_linear_norm(self.down_proj, act_q, residual, next_norm_w, quantize=next_quant)
                if h_next is not None:
                    return None, residual, h_next
            x = self.down_proj.forward(None, pre_quant=act_q)
        else:
            act = torch.empty(gate_up.size(0), self.I, dtype=gate_up.dtype, device=gate_up.device)
            ops.act_and_mul(act, gate_up, "silu")
            x = self.down_proj.forward(act)
        return x, residual, None

    def _gate_up_act_16(self, h):
        """unquantised layer: gate_up_proj with SiLU * mul in the GEMM epilogue (ops.matmul_silu_mul, packed 16-bit weights,
        decode shapes); None = not applicable"""
        lin = self.gate_up_proj       # (column-parallel: a rank's shard holds its gate rows, then its up rows -- no reduce)
        if lin.mode in ("int8", "fp8") or lin.weight_packed is None or h.dim() != 2:
            return None
        return ops.matmul_silu_mul(h, lin.weight, lin.bias, b_packed=lin.weight_packed)


class Qwen2Model:
    """layer loop + final norm + lm_head (llm_model_base.h:60-125, 193-204); lm_head is never quantised
    (linear.cpp:512-520) and column-parallel with gather_output (linear.cpp:712-714)."""

    def __init__(self, args: ModelArgs, mode: str = "int8", dtype=torch.bfloat16, device="cuda", seed: int = 0,
                 tp: Optional[parallel.ProcessGroup] = None, fuse: bool = True, n_layers: Optional[int] = None):
        gen = torch.Generator(device=device).manual_seed(seed)   # the SAME stream on every rank: full tensors, then shards
        self.args, self.tp, self.dtype, self.device = args, tp, dtype, device
        tp_size = tp.world_size() if tp else 1
        self.layers = [Qwen2DecoderLayer(args, mode, dtype, device, gen, tp, fuse)
                       for _ in range(n_layers if n_layers is not None else args.n_layers)]
        self.norm_w = (torch.rand(args.hidden_size, device=device, generator=gen) + 0.5).to(dtype)
        self.lm_head = QuantLinear(args.vocab_size, args.hidden_size, False, "16bit", dtype, device, gen,
                                   shard=("col", tp.rank() if tp else 0, tp_size),
                                   pack16=True)
        self.embed = (torch.randn(args.vocab_size, args.hidden_size, device=device, generator=gen)).to(dtype)
        self.cos_sin = build_cos_sin_cache(args, dtype, device, 8192)

    def forward(self, tokens, positions, md: AttentionMetadata, kv_caches):
        x = torch.nn.functional.embedding(tokens, self.embed)  # embed_tokens_ (llm_model_base.h:74)
        residual, h_in = None, None
        n = len(self.layers)
        for i, (layer, kvc) in enumerate(zip(self.layers, kv_caches)):
            last = i + 1 == n
            nxt = self.norm_w if last else self.layers[i + 1].input_norm_w
            x, residual, h_in = layer.forward(x, residual, positions, md, kvc, self.cos_sin, h_in=h_in, next_norm_w=nxt,
                                              next_quant=not last)
        if h_in is not None:
            return h_in      # the last layer's down_proj already produced the final norm (16-bit)
        if residual is None:
            out = torch.empty_like(x)
            ops.rms_norm(out, x, self.norm_w, self.args.rms_norm_eps)
        else:
            ops.fused_add_rms_norm(x, residual, self.norm_w, self.args.rms_norm_eps)
            out = x
        return out

    def logits(self, hidden):
        y = self.lm_head.forward(hidden)
        return parallel.gather(y, self.tp)

    def greedy_tokens(self, hidden):
        """lm_head -> Sampler::greedy_sample (argmax(-1), sampler.cpp:160-168) as ONE pass: the packed lm_head GEMM reduces its
        columns to (max logit, first index) per row in its epilogue and the [B, V] logits are never written
        (ops.matmul_argmax, round 4). Tensor parallel: every rank reduces its own column shard and the ranks exchange [B] (value,
        global index) pairs instead of all-gathering [B, V / tp] logits (linear.cpp:712-714) -- argmax commutes with the gather;
        ties go to the lowest global index, i.e. torch.argmax of the gathered logits. Token ids int64 [B], bit-equal to
        greedy_argmax(logits(hidden)). Falls back to exactly that outside the packed kernel's envelope. Policy
        (ops._GREEDY_FUSION): on by default under tensor parallelism only -- on one GPU the fused form TIES with the two operators
        (the GEMM is not store-bound: 333 vs 326 us stand-alone, +0.01 ms in the step; profiles/r04_lm_head_ab.txt)."""
        lm = self.lm_head
        tp_size = self.tp.world_size() if self.tp else 1
        fuse = ops._GREEDY_FUSION == "1" or (ops._GREEDY_FUSION == "auto" and tp_size > 1)
        if fuse and lm.mode == "16bit" and lm.weight_packed is not None and hidden.dim() == 2:
            got = ops.matmul_argmax(hidden, lm.weight_packed, lm.weight.size(0), lm.bias, want_value=tp_size > 1)
            if got is not None:
                if tp_size == 1:
                    return got
                idx, val = got
                return parallel.argmax_merge(val, idx + self.tp.rank() * lm.weight.size(0), self.tp)   # shard-local -> global column
        return ops.greedy_argmax(self.logits(hidden))


def to_deepseek_rope_layout(t: torch.Tensor) -> torch.Tensor:
    """deepseek_v2_attention.cpp:35-46: [.., d] viewed as [.., d/2, 2], transposed to [.., 2, d/2]: even dims first"""
    shape = t.shape
    return t.reshape(*shape[:-1], shape[-1] // 2, 2).transpose(-1, -2).reshape(shape).contiguous()


class DeepseekV2Attention:
    """DeepseekV2AttentionImpl (layers/dcu/deepseek_v2_attention.cpp:50-317), MLA in the absorbed form: the cache holds
    one latent row [rms_norm(c_kv) (kv_lora) || rope(k_pe) (rope)] per token; q_nope is absorbed by w_kc, scores run over
    the (kv_lora + rope)-dim latent, values are its first kv_lora dims, then bmm(w_vc) and o_proj (whose TP reduction
    is left to the decoder layer, enable_result_reduction = false). The projections are 16-bit here (the reference
    quantises q_b_proj / o_proj through QuantArgs; kv_a / kv_b / q_a are never quantised, :75-118).
    Differences from the reference, none visible in the result: prefill reads the latent rows back from the paged cache
    it has just written (mla_prefill) instead of looping over sequences on the host with torch SDPA (:212-262), so the
    layer has no host sync; chunked prefill works the same way (the reference CHECK-fails on it, :270-271)."""

    def __init__(self, hidden: int, n_heads: int, q_lora: int, kv_lora: int, nope: int, rope: int, v_dim: int, eps: float,
                 dtype, device, gen, tp: Optional[parallel.ProcessGroup] = None, rope_theta: float = 1e4,
                 max_pos: int = 8192, mscale: float = 1.0, quant: str = "16bit"):
        tp_size = tp.world_size() if tp is not None else 1
        self.quant = quant   # "fp8": q_b_proj and o_proj carry QuantArgs (deepseek_v2_attention.cpp:96-118); the others never do
        assert n_heads % tp_size == 0, "num_heads must be divisible by tensor parallel size"   # :64-65
        self.h, self.q_lora, self.kv_lora, self.nope, self.rope, self.v_dim = n_heads // tp_size, q_lora, kv_lora, nope, rope, v_dim
        self.eps, self.dtype, self.tp = eps, dtype, tp
        rnd = lambda n, k: (torch.randn(n, k, device=device, generator=gen) / math.sqrt(k)).to(dtype)
        ones = lambda n: (torch.rand(n, device=device, generator=gen) + 0.5).to(dtype)
        if q_lora > 0:
            self.q_a_w, self.q_a_norm_w, self.q_b_w = rnd(q_lora, hidden), ones(q_lora), rnd(self.h * (nope + rope), q_lora)
        else:
            self.q_w = rnd(self.h * (nope + rope), hidden)
        self.kv_a_w, self.kv_a_norm_w = rnd(kv_lora + rope, hidden), ones(kv_lora)
        kv_b = rnd(self.h * (nope + v_dim), kv_lora).unflatten(0, (self.h, nope + v_dim))
        # ONE orientation of each absorbed weight (the reference's w_kc_ [h, nope, kv_lora] / w_vc_ [h, kv_lora, v], load_state_dict
        # :335-339), stored the way ops.bmm_heads reads them (round 5): every head's matrix K-contiguous per output column,
        # [h, N, K]. For W_vc that is the slice of kv_b_proj's weight itself (a VIEW); W_kc is transposed once here.
        self.w_kc_nk = kv_b[:, :nope].transpose(1, 2).contiguous()    # [h, kv_lora, nope] = w_kc^T per head
        self.w_vc_nk = kv_b[:, nope:]                                 # [h, v, kv_lora] = w_vc^T per head (strided over h)
        self.o_w = rnd(hidden, self.h * v_dim)
        if quant != "16bit":
            assert q_lora > 0 and quant == "fp8"
            self.q_b_lin = QuantLinear(self.h * (nope + rope), q_lora, False, quant, dtype, device, gen)
            self.o_lin = QuantLinear(hidden, self.h * v_dim, False, quant, dtype, device, gen)
            self.q_b_w = self.o_w = None
        self.scale = float((nope + rope) ** -0.5) * mscale * mscale   # :148-154
        inv_freq = 1.0 / torch.pow(torch.tensor(rope_theta), torch.arange(0, rope, 2, dtype=torch.float32) / rope)
        fr = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq)
        self.cos_sin = torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(device)

    def _norm(self, x, w):
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        ops.rms_norm(out, x, w, self.eps)
        return out

    def forward(self, positions, hidden_states, md: AttentionMetadata, kv_cache: KVCache):
        T = hidden_states.size(0)
        assert positions.numel() == T, "position/token mismatch"                                  # :272-279
        latent = ops.matmul(hidden_states, self.kv_a_w)                                           # [T, kv_lora + rope]
        c_kv_normed = self._norm(latent[:, :self.kv_lora], self.kv_a_norm_w)
        k_pe = to_deepseek_rope_layout(latent[:, self.kv_lora:].contiguous().unsqueeze(1))      # [T, 1, rope]
        ops.rotary_embedding(positions, k_pe, None, self.cos_sin, True, head_size=self.rope)
        latent_normed = torch.cat([c_kv_normed, k_pe.squeeze(1)], -1)
        ops.store_latent_cache(latent_normed, md.slot_mapping, kv_cache.get_k_cache())
        if self.q_lora > 0:                                                                       # prepare_query :156-168
            q_a = self._norm(ops.matmul(hidden_states, self.q_a_w), self.q_a_norm_w)
            q = self.q_b_lin.forward(q_a) if self.quant != "16bit" else ops.matmul(q_a, self.q_b_w)
        else:
            q = ops.matmul(hidden_states, self.q_w)
        q = q.view(T, self.h, self.nope + self.rope)
        q_pe = to_deepseek_rope_layout(q[..., self.nope:].contiguous())
        ops.rotary_embedding(positions, q_pe, None, self.cos_sin, True, head_size=self.rope)
        # q_nope x W_kc (:310-311: torch::bmm over transposed views = rocBLAS in the reference): one launch of this backend's per-head
        # GEMM, reading the q_nope slice of the packed q tensor in place and writing the first kv_lora columns of the kernel's input
        q_in = torch.empty(T, self.h, self.kv_lora + self.rope, dtype=q.dtype, device=q.device)
        ops.bmm_heads(q[..., :self.nope], self.w_kc_nk, out=q_in[..., :self.kv_lora])             # [T, h, kv_lora]
        q_in[..., self.kv_lora:] = q_pe
        if md.is_prefill or md.is_chunked_prefill:
            attn = ops.mla_prefill(q_in, kv_cache.get_k_cache(), md.q_cu_seq_lens, md.kv_seq_lens, md.block_table,
                                   self.kv_lora, self.scale, md.max_seq_len, True)
        else:
            attn = ops.mla_decode(q_in, kv_cache.get_k_cache(), md.kv_seq_lens, md.block_table, self.kv_lora, self.scale,
                                  md.max_seq_len)
        out = ops.bmm_heads(attn.view(T, self.h, self.kv_lora), self.w_vc_nk).flatten(1, 2)       # project_output :180-187
        if self.quant != "16bit":
            return self.o_lin.forward(out)
        return ops.matmul(out, self.o_w)   # partial sums under TP: the decoder layer reduces


class FusedMoE:
    """FusedMoEImpl (layers/dcu/fused_moe.cpp:143-337), one expert-parallel rank (all experts local), bf16 like the
    reference's DCU path: select_experts = fused gating top-k + index build, expand, grouped GEMM w13
    [E, 2 * I / tp, H], SiLU * mul, grouped GEMM w2 [E, H, I / tp], weighted combine, all-reduce over TP.
    Differences from the reference, none of them visible in the result: the expanded activations and the un-sorted
    second GEMM output are never materialised (group_gemm_gather / moe_combine_sorted fuse the index_select and the
    index_copy_), and nothing reads a device tensor on the host, so the whole layer is graph-capturable."""

    def __init__(self, hidden: int, inter: int, n_experts: int, topk: int, dtype, device, gen, renormalize: bool = True,
                 scoring_func: str = "softmax", correction_bias=None, tp: Optional[parallel.ProcessGroup] = None,
                 fuse: bool = True, mode: str = "16bit", num_expert_group: int = 1, topk_group: int = 1,
                 route_scale: float = 1.0, ep: Optional[parallel.ProcessGroup] = None, ep_rank: Optional[int] = None,
                 ep_size: Optional[int] = None, shared_experts=None):
        self.E, self.topk, self.renorm, self.scoring, self.bias, self.tp, self.fuse = (
            n_experts, topk, renormalize, scoring_func, correction_bias, tp, fuse)
        # expert parallelism as in the reference's DCU path (fused_moe.cpp:53-63, 236-315): every rank sees every token and
        # routes over ALL experts, computes only experts [start, start + E / ep), and the EP all-reduce adds the ranks.
        # (ep_rank / ep_size without a group: one rank's share computed alone, used by the single-GPU tests.)
        self.ep = ep
        self.ep_size = ep_size if ep_size is not None else (ep.world_size() if ep is not None else 1)
        self.ep_rank = ep_rank if ep_rank is not None else (ep.rank() if ep is not None else 0)
        assert n_experts % self.ep_size == 0
        self.E_local = n_experts // self.ep_size
        self.start = self.ep_rank * self.E_local
        self.shared = shared_experts   # callable [T, H] -> [T, H] (the shared experts' dense MLP), added after the reduces
        self._side = None
        # DeepSeek-style device-limited routing (fused_moe.cpp:155-166: num_expert_group / topk_group / route_scale)
        self.n_group, self.topk_group, self.route_scale = num_expert_group, topk_group, route_scale
        self.mode = mode  # "16bit" = the reference's DCU path; "int8" = W8A8 experts (GroupGemmParams a_scale / b_scale)
        tp_size = tp.world_size() if tp is not None else 1
        assert inter % tp_size == 0
        i_local = inter // tp_size
        self.w13 = (torch.randn(n_experts, 2 * i_local, hidden, device=device, generator=gen) / math.sqrt(hidden)).to(dtype)
        self.w2 = (torch.randn(n_experts, hidden, i_local, device=device, generator=gen) / math.sqrt(inter)).to(dtype)
        if self.ep_size > 1:  # every rank draws the same full tensors and keeps its own experts
            self.w13 = self.w13[self.start:self.start + self.E_local].contiguous()
            self.w2 = self.w2[self.start:self.start + self.E_local].contiguous()
        if mode == "int8":  # symmetric per-output-channel int8 (what a W8A8 checkpoint carries)
            def q8(w):
                sc = (w.float().abs().amax(-1) / 127.0).clamp_min(1e-12)
                return torch.round(w.float() / sc[..., None]).clamp_(-127, 127).to(torch.int8), sc
            self.w13_q, self.w13_s = q8(self.w13)
            self.w2_q, self.w2_s = q8(self.w2)

    def forward_experts(self, hidden_states, router_logits):
        x = hidden_states.reshape(-1, hidden_states.size(-1))
        T = x.size(0)
        shared_out, side = None, None
        if self.shared is not None and x.is_cuda and not torch.cuda.is_current_stream_capturing():
            # shared experts on a second stream next to the routed path (fused_moe.cpp:304-335)
            if self._side is None:
                self._side = torch.cuda.Stream(device=x.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                shared_out = self.shared(x)
        weights, ids = ops.moe_active_topk(router_logits.reshape(T, -1), self.topk, self.n_group, self.topk_group,
                                           self.renorm, self.bias, self.scoring, self.route_scale)
        local = None
        if self.ep_size > 1:
            # rotate the expert ids so that this rank's experts sort to the front: their rows are then the first
            # sum(sizes[:E_local]) sorted rows, at offset 0 -- no host read of the sizes (the reference's .item() calls)
            ids = torch.remainder(ids - self.start, self.E).to(torch.int32)
        src_dst, dst_src, sizes = ops.moe_compute_index(ids, self.E)
        if self.ep_size > 1:
            sizes = sizes[:self.E_local]
            local = sizes
        if self.mode == "int8":
            # each token is quantised ONCE; the expand happens inside the first grouped GEMM's A staging (scales follow)
            xq, xs = ops.scaled_quantize(x)
            h13 = ops.group_gemm_w8a8(xq, xs, self.w13_q, self.w13_s, sizes, x.dtype, row_index=dst_src, index_div=self.topk)
            aq, a_s = ops.act_and_mul_dynamic_int8_quant(h13, "silu", live_sizes=local)   # EP: the rank's own rows only
            h2 = ops.group_gemm_w8a8(aq, a_s, self.w2_q, self.w2_s, sizes, x.dtype)
            out = ops.moe_combine_sorted(h2, src_dst, weights, T, self.topk, local)
        else:
            h13 = ops.group_gemm_gather(x, dst_src, self.topk, self.w13, sizes) if self.fuse else None
            if h13 is None:  # reference order: expand with index_select, then the grouped GEMM
                h13 = ops.group_gemm(x.index_select(0, (dst_src // self.topk).long()), self.w13, sizes)
            act = torch.empty(h13.size(0), h13.size(1) // 2, dtype=h13.dtype, device=h13.device)
            ops.act_and_mul(act, h13, "silu")
            h2 = ops.group_gemm(act, self.w2, sizes)
            if self.fuse:
                out = ops.moe_combine_sorted(h2, src_dst, weights, T, self.topk, local)
            else:  # the reference's own sequence: zeros, index_copy_ of the local rows (host read of their count), combine
                n_local = int(sizes.sum().item()) if local is not None else h2.size(0)
                full = torch.zeros_like(h2) if local is not None else torch.empty_like(h2)
                full.index_copy_(0, dst_src[:n_local].long(), h2[:n_local])
                out = ops.moe_combine_result(full, weights, T, self.topk)
        out = parallel.reduce(out, self.ep)
        if self.shared is not None and side is None:
            # no second stream (graph capture / CPU): the TP all-reduce of the routed output is started first and the shared
            # experts run while it is in flight (the reference's launch_reduce / finish_reduce pair,
            # deepseek_v2_sparse_moe_block.cpp:208-260)
            pending = parallel.launch_reduce(out, self.tp)
            shared_out = self.shared(x)
            out = parallel.finish_reduce(pending)
        else:
            out = parallel.reduce(out, self.tp)
        if self.shared is not None:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            out = out + shared_out
        return out.reshape(hidden_states.shape)


    def forward_experts_alltoall(self, hidden_states, router_logits):
        """all-to-all expert parallelism (parallel.ep_dispatch / ep_combine; DeepEP-style, N4): this rank's OWN tokens are
        routed over all experts, every (token, k) row travels to the rank that owns its expert, is computed there by the
        same grouped GEMMs, and travels back for the weighted combine. Not graph-capturable (split sizes on the host)."""
        x = hidden_states.reshape(-1, hidden_states.size(-1))
        T = x.size(0)
        pg = self.ep if self.ep is not None else parallel.ProcessGroup(None, 0, 1)
        assert pg.world_size() == self.ep_size
        weights, ids = ops.moe_active_topk(router_logits.reshape(T, -1), self.topk, self.n_group, self.topk_group,
                                           self.renorm, self.bias, self.scoring, self.route_scale)
        rows, local_e, ctx = parallel.ep_dispatch(x, ids, self.E, pg)
        R = rows.size(0)
        if R > 0:
            src_dst, dst_src, sizes = ops.moe_compute_index(local_e.view(R, 1), self.E_local)
            if self.mode == "int8":
                xq, xs = ops.scaled_quantize(rows)
                h13 = ops.group_gemm_w8a8(xq, xs, self.w13_q, self.w13_s, sizes, x.dtype, row_index=dst_src, index_div=1)
                aq, a_s = ops.act_and_mul_dynamic_int8_quant(h13, "silu")
                h2 = ops.group_gemm_w8a8(aq, a_s, self.w2_q, self.w2_s, sizes, x.dtype)
            else:
                h13 = ops.group_gemm_gather(rows, dst_src, 1, self.w13, sizes)
                if h13 is None:
                    h13 = ops.group_gemm(rows.index_select(0, dst_src.long()), self.w13, sizes)
                act = torch.empty(h13.size(0), h13.size(1) // 2, dtype=h13.dtype, device=h13.device)
                ops.act_and_mul(act, h13, "silu")
                h2 = ops.group_gemm(act, self.w2, sizes)
            y = h2.index_select(0, src_dst.long())      # expert order -> arrival order
        else:
            y = rows.new_empty(0, x.size(-1))
        back = parallel.ep_combine(y, ctx, pg)
        out = ops.moe_combine_result(back, weights, T, self.topk)
        out = parallel.reduce(out, self.tp)
        if self.shared is not None:
            out = out + self.shared(x)
        return out.reshape(hidden_states.shape)


class DualBatchDecoder:
    """Decode step of two micro-batches on two HIP streams (the reference's enable_multi_stream_parallel with
    micro_batch_num = 2, framework/config/parallel_config.h:83-85). The paged-attention kernel of a layer is bound by
    HBM and leaves the matrix pipes, the LDS and most of the power budget idle; the linear layers are bound by exactly
    those. So the attention kernels of the two halves are chained back to back (A0, B0, A1, B1, ...) and everything
    else of a half (o_proj, MLP, next qkv_proj, RoPE + KV write) runs on that half's own stream underneath the other
    half's attention. Arithmetic per sequence is unchanged (row-wise kernels and GEMM rows are independent, the
    attention kernel handles sequences independently), so the logits equal the single-batch step bit for bit.
    W8A8 fused decode path. Tensor parallel (round 4): allowed; each half's row-parallel linears reduce through the group's own
    all-reduce (RCCL on its own stream) and the one-shot kernel is SUSPENDED for the duration of forward() -- also under graph
    capture -- because its launches from two streams of one rank would share one epoch / flag / slot state (round-4 advisor)."""

    def __init__(self, model: "Qwen2Model", md: AttentionMetadata, batch: int):
        # tensor parallel (round 4): allowed -- each half's row-parallel linears reduce through the group's own all-reduce (RCCL runs
        # it on its own stream, fenced against the half's stream), which is what lets the OTHER half's GEMMs overlap the transfer.
        # The one-shot kernel keeps ONE epoch / flag / slot state per rank: forward() suspends it explicitly (its "one stream" rule
        # accepts any stream while a capture is running, which would let both forked branches launch it).
        assert all(l.fuse for l in model.layers)
        self.model, self.B = model, batch
        h = batch // 2
        self.cut = [(0, h), (h, batch)]
        self.md = [AttentionMetadata(
            q_cu_seq_lens=torch.arange(e - b + 1, dtype=torch.int32, device=md.kv_seq_lens.device),
            kv_cu_seq_lens=None, kv_seq_lens=md.kv_seq_lens[b:e].contiguous(), slot_mapping=md.slot_mapping[b:e].contiguous(),
            block_table=md.block_table[b:e].contiguous(), max_query_len=1, max_seq_len=md.max_seq_len) for b, e in self.cut]
        self.streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        self.chain = True      # (tools set .chain = False to measure the halves without the cross-stream hand-over)
        for st, (b, e) in zip(self.streams, self.cut):   # split-K partial sums of the two halves must not mix
            ops.set_gemm_workspace_for_stream(st, 32 << 20)

    def close(self):
        """give the per-stream scratch buffers back (the C side keeps at most 8 registered streams)"""
        for st in self.streams:
            ops.release_stream_workspaces(st)
        self.streams = []

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def forward(self, tokens, positions, kv_caches):
        oneshot = self.model.tp.oneshot if self.model.tp is not None else None
        if oneshot is not None:
            with oneshot.suspended():
                return self._forward(tokens, positions, kv_caches)
        return self._forward(tokens, positions, kv_caches)

    def _forward(self, tokens, positions, kv_caches):
        m = self.model
        main = torch.cuda.current_stream()
        x_full = torch.nn.functional.embedding(tokens, m.embed)
        out = torch.empty_like(x_full)
        n = len(m.layers)
        state = []
        for i, ((b, e), st) in enumerate(zip(self.cut, self.streams)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                q, res = m.layers[0].pre_attention(x_full[b:e], None, positions[b:e], self.md[i], kv_caches[0], m.cos_sin)
            state.append([q, res])
        prev_attn_done = None
        for li, layer in enumerate(m.layers):
            last = li + 1 == n
            nxt = m.norm_w if last else m.layers[li + 1].input_norm_w
            for i, ((b, e), st) in enumerate(zip(self.cut, self.streams)):
                with torch.cuda.stream(st):
                    if prev_attn_done is not None and self.chain:
                        # attention kernels of the two halves never overlap. The dependency is routed through the
                        # origin stream: ROCm 7.2 segfaults in hipStreamEndCapture when a forked stream waits on an
                        # event recorded by another forked stream.
                        main.wait_event(prev_attn_done)
                        hub = torch.cuda.Event()
                        hub.record(main)
                        st.wait_event(hub)
                    o_in = layer.attention_kernel(state[i][0], self.md[i], kv_caches[li])
                    prev_attn_done = torch.cuda.Event()
                    prev_attn_done.record(st)
                    x, res, h_next = layer.post_attention(o_in, state[i][1], nxt, not last)
                    if last:
                        if h_next is None:   # down_proj was not fused with the final norm
                            ops.fused_add_rms_norm(x, res, m.norm_w, m.args.rms_norm_eps)
                            h_next = x
                        out[b:e].copy_(h_next)
                    else:
                        q, res = m.layers[li + 1].pre_attention(x, res, positions[b:e], self.md[i], kv_caches[li + 1],
                                                               m.cos_sin, h_in=h_next)
                        state[i] = [q, res]
        for st in self.streams:
            main.wait_stream(st)
        return out
