"""bench_slices.py -- the single-GPU slices of BASELINE.json's TP=8 configurations (bench.py --config cfg4-slice | cfg5-slice).

configs[3] (DeepSeek-V3 MLA fp8 TP=8, bs=128 ctx=8192) and configs[4] (Qwen3-MoE W8A8 TP=8, T=8192 chunked prefill) do not
fit the metric's one-GPU line; what one rank of those jobs runs per layer does, and that is what is timed here, with the shard
shapes of rank 0 of 8 and no exchange (the exchanges are counted and reported, not timed -- a 1-GPU box has no peer):

  cfg4-slice  one DeepSeek-V3 decoder layer of one TP=8 rank, decode step of 128 sequences with 8191 cached tokens each:
              DeepseekV2Attention (16 of 128 heads, q_lora 1536, kv_lora 512, rope 64; q_b_proj / o_proj fp8 e4m3 with dynamic
              per-token activation scales; paged latent cache [n_blocks, 64, 1, 576] bf16, shuffled pages) + the routed experts
              (256 experts, top-8 of 4-of-8 groups, sigmoid scores + correction bias, intermediate 2048 / 8 per rank, 16-bit
              like the reference's DCU path, fused_moe.cpp:217-337).  roofline: mla_decode kernel against HBM.
  cfg5-slice  one Qwen3-MoE layer on one rank of 8: the attention block (4 of 32 q heads, W8A8 qkv / o, per-head q/k RMSNorm +
              RoPE, chunked prefill through the block table) and the routed experts (H 2048, 128 experts top-8, intermediate 768) on one expert-parallel
              rank of 8 (16 local experts; every rank routes all T = 8192 tokens of the chunk, fused_moe.cpp:236-315), W8A8:
              per-token int8 quantisation, grouped GEMM w13 with the expand fused in, SiLU*mul + requantisation, grouped GEMM
              w2, weighted combine.  roofline: the w13 grouped GEMM against the dense int8 MFMA rate.

The product path only (xllm_amd.layers over the C ABI); nothing under oracle/ is touched.
"""
import time

import torch

HBM_PEAK_GBS = 8000.0
I8_PEAK_TOPS = 5000.0   # dense int8 MFMA (2x the bf16 rate), MI355X_MICROARCH.md


def _timed_calls(fn_holder, name, events):
    """wrap module attribute `name` so that every call is bracketed by HIP events on the launch stream"""
    orig = getattr(fn_holder, name)

    def wrapped(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*args, **kw)
        e1.record()
        events.append((e0, e1))
        return out

    setattr(fn_holder, name, wrapped)
    return orig


def _time_steps(step, steps, warmup, sync):
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps


def cfg4_slice(a, dev):
    from xllm_amd import attention, layers, ops
    from xllm_amd.attention import KVCache
    H, heads, tp = 7168, 128, 8
    q_lora, kv_lora, nope, rope, v_dim = 1536, 512, 128, 64, 128
    E, topk, n_group, topk_group, moe_i = 256, 8, 8, 4, 2048
    B, ctx, bs = 128, 8192, 64
    gen = torch.Generator(device=dev).manual_seed(4)
    attn = layers.DeepseekV2Attention(H, heads // tp, q_lora, kv_lora, nope, rope, v_dim, 1e-6, torch.bfloat16, dev, gen,
                                      max_pos=ctx, quant="fp8")
    bias = torch.randn(E, device=dev, generator=gen) * 0.1
    moe = layers.FusedMoE(H, moe_i // tp, E, topk, torch.bfloat16, dev, gen, renormalize=True, scoring_func="sigmoid",
                          correction_bias=bias, num_expert_group=n_group, topk_group=topk_group, route_scale=2.5)
    gate_w = (torch.randn(E, H, device=dev, generator=gen) / H ** 0.5).bfloat16()
    pages = ctx // bs
    n_blocks = int(B * pages * 1.1) + 1
    perm = torch.randperm(n_blocks, generator=torch.Generator().manual_seed(5))[: B * pages].to(torch.int32).view(B, pages)
    bi = attention.build_batch_input([ctx - 1] * B, [ctx] * B, perm.tolist(), bs)
    md = attention.build_attention_metadata(bi, is_prefill=False, is_chunked_prefill=False, device=dev)
    cache = KVCache(torch.empty(n_blocks, bs, 1, kv_lora + rope, dtype=torch.bfloat16, device=dev).normal_(generator=gen), None)
    x = torch.randn(B, H, device=dev, generator=gen).bfloat16()
    pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)

    # post_attention_layernorm (deepseek_v2 decoder layer: residual add + RMSNorm between the attention and the MoE block). Round 6:
    # without it the gate saw the raw attention output (|h| ~ 0.04 at random init), the correction bias (sigma 0.1) decided the
    # routing and all 128 tokens chose the same ~10 experts -- 128 rows per expert, eight 16-row passes over their weights, nothing
    # like a decode step's ~4 rows on (nearly) all 256 experts. With the norm the logits have sigma ~ 1 and the routing is balanced.
    post_w = torch.ones(H, device=dev, dtype=torch.bfloat16)
    resid = x.clone()
    route_seen = []

    def step():
        h = attn.forward(pos, x, md, cache)                 # (all-reduce over the 8 ranks here)
        r = resid.clone()
        ops.fused_add_rms_norm(h, r, post_w, 1e-6)          # h <- RMSNorm(h + residual)
        logits = ops.matmul(h, gate_w)
        if not route_seen:
            route_seen.append(logits)
        return moe.forward_experts(h, logits)               # (all-reduce over the 8 ranks here)

    sync = torch.cuda.synchronize
    step()
    sync()
    _w, _ids = ops.moe_active_topk(route_seen[0].reshape(B, -1), topk, n_group, topk_group, True, bias, "sigmoid", 2.5)
    _cnt = torch.bincount(_ids.flatten().long(), minlength=E)
    routing = {"experts_with_rows": int((_cnt > 0).sum()), "max_rows_per_expert": int(_cnt.max()), "rows": int(_cnt.sum())}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    dt = _time_steps(g.replay, a.steps, a.warmup, sync)
    # dominant kernel, per launch, eagerly on the same stream with events (a replayed graph has no host-visible launches)
    ev, fp8_ev = [], []
    orig = _timed_calls(ops, "mla_decode", ev)
    orig8 = _timed_calls(ops, "fp8_scaled_matmul", fp8_ev)
    for _ in range(max(a.steps, 5)):
        step()
    sync()
    ops.mla_decode, ops.fp8_scaled_matmul = orig, orig8
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    attn_ms = sum(ms) / len(ms)
    h_l = heads // tp
    nbytes = B * (ctx * (kv_lora + rope) * 2 + h_l * (kv_lora + rope) * 2 + h_l * kv_lora * 2)
    achieved = nbytes / (attn_ms * 1e-3) / 1e9
    f8 = [e0.elapsed_time(e1) for e0, e1 in fp8_ev]
    qb_eager_ms, o_eager_ms = sum(f8[0::2]) / len(f8[0::2]), sum(f8[1::2]) / len(f8[1::2])

    def graph_time(lin, n=100):
        """the linear alone (dynamic fp8 quant is timed separately upstream): n launches in one HIP graph, two timed replays --
        eager HIP events around a 10-us launch mostly measure the host (round 3)"""
        xin = (torch.randn(B, lin.weight.size(1), device=dev, generator=gen) * 2).to(torch.float8_e4m3fn)
        xs = torch.full((1,), 0.03, device=dev)
        fn = lambda: ops.fp8_scaled_matmul(xin, lin.weight, xs, lin.w_scale, lin.dtype, lin.bias, b_packed=lin.weight_packed)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, stream=st):
                for _ in range(n):
                    fn()
            gg.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gg.replay()
            gg.replay()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * n)

    qb_ms, o_ms = graph_time(attn.q_b_lin), graph_time(attn.o_lin)
    # HBM bytes of the dominant kernel from the PMC counters, in this run (two rocprofv3 --pmc passes over tools/mla_pmc_probe.py: the
    # same kernel at the same shape in a child process; the guide's gfx950 correction is applied by bench.live_pmc_traffic)
    traffic, traffic_source = None, None
    if not getattr(a, "no_pmc", False):
        import bench as _bench
        traffic, traffic_source = _bench.live_pmc_traffic(B, ctx, h_l, 1, probe_name="mla_pmc_probe.py", kernel_like="%mla_decode%",
                                                          probe_args=(B, ctx, h_l, 4))
    qb_bytes, o_bytes = h_l * (nope + rope) * q_lora, H * h_l * v_dim
    return {
        "metric": "decode tokens/s through one DeepSeek-V3 layer of one TP=8 rank (cfg4-slice)",
        "value": round(B / dt, 1), "unit": "tokens/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "fp8",
        "config": {"workload": "deepseek_v3 layer slice: MLA decode (16 heads/rank, fp8 q_b/o projections) + residual add / RMSNorm + "
                               "256-expert top-8 routed MoE (intermediate 2048/8), bs=128 ctx=8192, paged latent cache block=64 bf16, "
                               "random-init weights", "global_batch": B, "ctx": ctx, "per_gpu_batch": B,
                   "parallelism": "rank 0 of tp8 (shard shapes, no exchange timed)", "collectives_per_step": 2,
                   "hip_graph": True, "routing": routing,
                   # round 5: the two absorbed-weight batch GEMMs (q_nope x w_kc, attn x w_vc; torch::bmm = rocBLAS in the reference,
                   # deepseek_v2_attention.cpp:180-187, 310-311) run on this backend's per-head GEMM (xllm_mi355_bmm_heads), in place
                   # on the token-major tensors; what is left of the vendor library on the measured path is copy / embedding glue
                   "vendor_library_ops": ["torch copy / embedding glue"]},
        "roofline": {"bound": "hbm", "kernel": "mla_decode", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "bytes_per_launch": nbytes, "avg_launch_ms": round(attn_ms, 4), "launches_timed": len(ms)},
        "fp8_linears": {"q_b_proj": {"M": B, "N": h_l * (nope + rope), "K": q_lora, "us": round(qb_ms * 1e3, 2),
                                      "weight_gbs": round(qb_bytes / (qb_ms * 1e-3) / 1e9, 1),
                                      "us_eager_events": round(qb_eager_ms * 1e3, 2)},
                        "o_proj": {"M": B, "N": H, "K": h_l * v_dim, "us": round(o_ms * 1e3, 2),
                                   "weight_gbs": round(o_bytes / (o_ms * 1e-3) / 1e9, 1),
                                   "us_eager_events": round(o_eager_ms * 1e3, 2)},
                        "kernel": "gemm_ws (packed e4m3 weights, v_mfma_f32_16x16x128_f8f6f4); us = per launch inside a "
                                  "100-launch HIP graph"},
    }


class _Qwen3AttentionRank:
    """the attention block of one Qwen3-MoE layer on one TP = 8 rank (qwen2_attention.cpp:132-193 with the Qwen3 branch :146-171:
    per-head RMSNorm of q and k + RoPE inside the packed qkv; 32 / 8 = 4 q heads, 4 kv heads replicated -> 1 per rank), W8A8
    linears, chunked prefill over the paged cache. The TP all-reduce after o_proj is counted, not timed."""

    def __init__(self, H, nq, nkv, d, dev, gen, max_pos):
        from xllm_amd import layers
        self.nq, self.nkv, self.d = nq, nkv, d
        self.qkv = layers.QuantLinear((nq + 2 * nkv) * d, H, False, "int8", torch.bfloat16, dev, gen)
        self.o = layers.QuantLinear(H, nq * d, False, "int8", torch.bfloat16, dev, gen)
        self.norm_w = (torch.rand(H, device=dev, generator=gen) + 0.5).bfloat16()
        self.q_w = (torch.rand(d, device=dev, generator=gen) + 0.5).bfloat16()
        self.k_w = (torch.rand(d, device=dev, generator=gen) + 0.5).bfloat16()
        args = layers.ModelArgs(H, 1, nq, nkv, d, 1, 1, 1e-6, 1e6, max_pos)
        self.cos_sin = layers.build_cos_sin_cache(args, torch.bfloat16, dev, max_pos)

    def forward(self, x, positions, md, cache):
        from xllm_amd import ops
        q8, s8 = ops.rms_norm_dynamic_int8_quant(x, self.norm_w, 1e-6)
        qkv = self.qkv.forward(None, pre_quant=(q8, s8))
        ops.fused_qk_norm_rope(qkv, self.nq, self.nkv, self.nkv, self.d, 1e-6, self.q_w, self.k_w, self.cos_sin, False, positions)
        qs, ks = self.nq * self.d, self.nkv * self.d
        q = qkv[:, :qs].unflatten(-1, (self.nq, self.d))
        k = qkv[:, qs:qs + ks].unflatten(-1, (self.nkv, self.d))
        v = qkv[:, qs + ks:].unflatten(-1, (self.nkv, self.d))
        ops.reshape_paged_cache(md.slot_mapping, k, v, cache.get_k_cache(), cache.get_v_cache())
        attn = ops.paged_attention(q, cache.get_k_cache(), cache.get_v_cache(), md.q_cu_seq_lens, md.kv_seq_lens, md.block_table,
                                   md.max_query_len, md.max_seq_len, self.d ** -0.5, True)
        return self.o.forward(attn)


def cfg5_slice(a, dev):
    from xllm_amd import attention, layers, ops
    from xllm_amd.attention import KVCache
    H, E, topk, moe_i, ep = 2048, 128, 8, 768, 8
    T, n_seq, bs = 8192, 2, 128
    gen = torch.Generator(device=dev).manual_seed(5)
    moe = layers.FusedMoE(H, moe_i, E, topk, torch.bfloat16, dev, gen, renormalize=True, mode="int8", ep_rank=0, ep_size=ep)
    gate_w = (torch.randn(E, H, device=dev, generator=gen) / H ** 0.5).bfloat16()
    x = torch.randn(T, H, device=dev, generator=gen).bfloat16()
    # attention of the chunk: 2 sequences x 4096 new tokens written to fresh pages and attended through the block table
    attn = _Qwen3AttentionRank(H, 32 // 8, 1, 128, dev, gen, T // n_seq)
    pages = T // n_seq // bs
    table = torch.randperm(n_seq * pages, generator=torch.Generator().manual_seed(6)).to(torch.int32).view(n_seq, pages)
    bi = attention.build_batch_input([0] * n_seq, [T // n_seq] * n_seq, table.tolist(), bs)
    md = attention.build_attention_metadata(bi, is_prefill=False, is_chunked_prefill=True, device=dev)
    cache = KVCache(torch.zeros(n_seq * pages, bs, 1, 128, dtype=torch.bfloat16, device=dev),
                    torch.zeros(n_seq * pages, bs, 1, 128, dtype=torch.bfloat16, device=dev))
    pos = bi.positions.to(dev).long()

    def step():
        h = x + attn.forward(x, pos, md, cache)                 # (TP all-reduce over the 8 ranks here)
        return moe.forward_experts(h, ops.matmul(h, gate_w))   # (EP all-reduce over the 8 ranks here)

    sync = torch.cuda.synchronize
    step()
    sync()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    dt = _time_steps(g.replay, a.steps, a.warmup, sync)
    ev = []
    orig = _timed_calls(ops, "group_gemm_w8a8", ev)
    sizes_seen = []
    orig_idx = ops.moe_compute_index

    def idx(*args, **kw):
        r = orig_idx(*args, **kw)
        sizes_seen.append(r[2])
        return r

    ops.moe_compute_index = idx
    for _ in range(max(a.steps, 5)):
        step()
    sync()
    ops.group_gemm_w8a8, ops.moe_compute_index = orig, orig_idx
    rows = int(sizes_seen[-1][:E // ep].sum().item())       # rows of this rank's 16 experts (host read AFTER the timed region)
    w13 = [e0.elapsed_time(e1) for e0, e1 in ev[0::2]]
    w2 = [e0.elapsed_time(e1) for e0, e1 in ev[1::2]]
    w13_ms, w2_ms = sum(w13) / len(w13), sum(w2) / len(w2)
    ops13 = 2.0 * rows * (2 * moe_i) * H
    ops2 = 2.0 * rows * H * moe_i
    ach = ops13 / (w13_ms * 1e-3) / 1e12
    return {
        "metric": "prefill tokens/s through the routed experts of one Qwen3-MoE layer on one EP=8 rank (cfg5-slice)",
        "value": round(T / dt, 1), "unit": "tokens/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "int8",
        "config": {"workload": "qwen3_moe layer slice: attention of one TP=8 rank (W8A8 qkv / o, per-head q/k RMSNorm + RoPE, KV write, "
                               "chunked prefill of 2 x 4096 tokens through the block table, 4 q heads / 1 kv head) + gate + top-8 "
                               "of 128 + W8A8 grouped GEMMs (16 local experts, intermediate 768) + combine, random-init weights",
                   "global_batch": T, "ctx": 4096, "per_gpu_batch": T, "rows_on_this_rank": rows,
                   "parallelism": "rank 0 of ep8 (all-reduce EP: every rank routes the whole chunk; no exchange timed)",
                   "collectives_per_step": 2, "hip_graph": True},
        "roofline": {"bound": "mfma", "kernel": "group_gemm_w8a8 (w13, expand fused)", "achieved": round(ach, 1),
                     "peak": I8_PEAK_TOPS, "unit": "TFLOP/s", "frac": round(ach / I8_PEAK_TOPS, 4), "traffic": None,
                     "ops_per_launch": ops13, "avg_launch_ms": round(w13_ms, 4), "launches_timed": len(w13)},
        "w2_gemm": {"tops": round(ops2 / (w2_ms * 1e-3) / 1e12, 1), "avg_launch_ms": round(w2_ms, 4)},
    }


def run(a, dev):
    out = (cfg4_slice if a.config == "cfg4-slice" else cfg5_slice)(a, dev)
    out.update({"n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "data": "synthetic"})
    return out
