#!/usr/bin/env python3
"""bench.py -- decode-step throughput of the xLLM hot path on MI355X (BASELINE.json metric).

A "step" is ONE decode iteration of the full Qwen2-7B W8A8 model (28 layers + final norm + lm_head + greedy
argmax) over a synthetic fixed-shape batch: bs=256 sequences with ctx=4096 cached tokens each, paged KV
(block 128, shuffled non-contiguous pages), inputs resident in HBM before the timed region.  Every layer runs
the reference's op order (xllm_amd/layers.py) through the C ABI of include/xllm_mi355.h -- hand-written
gfx950 kernels only; the oracle is used for the cpu_baseline leg alone.

Multi-GPU (one process per GPU; launched by torch.distributed.run, or by this script itself when WORLD_SIZE is not set):
the global batch of 256 sequences is FIXED ("scaling": "strong") and the layout decides how N GPUs share it:
  tp     (headline for N = 2, 4) the reference's tensor parallelism over xGMI (heads / columns per rank, SUM all-reduce after
         o_proj and down_proj, all-gather of logits; linear.cpp:1518-1520, 712-714); Qwen2-7B has 28 heads, so TP is 1, 2 or 4
         (qwen2_attention.cpp:54-65). The 56 per-layer sums run on the one-shot all-reduce kernel of csrc/allreduce.hip (self-tested
         at set-up, fused with the residual add + RMSNorm + int8 quant that follows; RCCL when the self-test fails --
         config.allreduce says which); greedy sampling reduces every rank's lm_head shard to [B] (max, index) pairs in the GEMM
         epilogue and exchanges THOSE instead of all-gathering the logits (round 4). `layouts` also carries the two other
         exchange designs of the same layout from the same launch: <layout>_rccl (the group's own all-reduce in stream, piecewise
         graphs) and <layout>_rccl_overlap (RCCL on its own stream under the dual micro-batch executor, eager);
  tp4dp2 (headline for N = 8) TP = 4 inside two replicas (what the reference would run on 8 GPUs);
  dp     N replicas of the model (7.6 GB of int8 weights fit a 288 GB GPU many times over), 256 / N sequences each, NO data-path
         exchange. A legitimate deployment answer for this path, measured in the same run and reported in `layouts` beside the
         headline (never substituted for it); --layout dp makes it the headline.

roofline.traffic (N = 1) is measured in the same run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the dominant kernel
at the bench's shape in a child process (tools/attn_pmc_probe.py; --no-pmc quotes the committed record instead).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (model, mode, global batch, ctx)
    "cfg3": ("qwen2_7b", "int8", 256, 4096),   # BASELINE.json metric config (configs[2])
    "cfg2": ("qwen2_7b", "16bit", 64, 2048),   # configs[1]
    "tiny": ("qwen2_0_5b", "int8", 8, 512),    # CPU-sized smoke shape
    # one rank's layer of the TP=8 configurations (configs[3], configs[4]) on one GPU: bench_slices.py
    "cfg4-slice": ("deepseek_v3", "fp8", 128, 8192),
    "cfg5-slice": ("qwen3_moe", "int8", 8192, 4096),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="cfg3", choices=list(CONFIGS))
    p.add_argument("--no-fuse", action="store_true", help="reference op order without the N1 quant fusions")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    p.add_argument("--graph", action="store_true",
                   help="multi-GPU: capture the RCCL collectives INTO one HIP graph (default for N>1 is the piecewise "
                        "replay of xllm_amd.parallel.PiecewiseGraph: kernels between collectives are graphs, the "
                        "collectives stay eager -- a failed capture of a collective cannot be recovered in-process)")
    p.add_argument("--dual", action="store_true",
                   help="decode the batch as two micro-batches on two streams (attention of one half under the linear "
                        "layers of the other; xllm_amd.layers.DualBatchDecoder), TP=1 only")
    p.add_argument("--micro", action="store_true", help="also print per-operator timings (stderr)")
    p.add_argument("--no-prefill", action="store_true", help="skip the prefill-TFLOPS leg")
    p.add_argument("--no-gemm", action="store_true", help="skip the quantised-GEMM leg (int8 / fp8 gate_up at M = 8192 and M = 128)")
    p.add_argument("--no-engine", action="store_true", help="skip the step-level harness leg (xllm_amd.engine.DecodeEngine)")
    p.add_argument("--layout", default="auto", choices=["auto", "dp", "tp", "tp4dp2"],
                   help="how N GPUs share the fixed global batch (auto = tp for N = 2, 4, tp4dp2 for N = 8, with the dp line "
                        "measured beside it; see the module docstring)")
    p.add_argument("--oneshot-allreduce", action="store_true",
                   help="(default since round 3; kept for old command lines) tensor parallel: the one-shot xGMI all-reduce of "
                        "csrc/allreduce.hip for the per-layer sums, self-tested at set-up, RCCL when the self-test fails")
    p.add_argument("--no-oneshot-allreduce", action="store_true", help="tensor parallel: RCCL for every collective")
    p.add_argument("--via-shim", action="store_true",
                   help="also run the SAME decode step through the C++ libtorch shim (shim/: the reference's operator signatures, "
                        "what a -DUSE_MI355 build of xLLM calls) in the reference's operator order, and print its ms_per_step "
                        "beside the headline")
    p.add_argument("--no-pmc", action="store_true",
                   help="skip the live rocprofv3 --pmc passes behind roofline.traffic (then the committed record is quoted)")
    p.add_argument("--no-per-rank", action="store_true",
                   help="N = 1: skip the per_rank_emulated block (one rank's step at the TP4 x DP2 and DP8 shard shapes, no exchange)")
    p.add_argument("--no-allocator-pages", action="store_true",
                   help="N = 1: skip the allocator_order_pages leg (the same step with the block table a fresh BlockManagerImpl pool "
                        "would hand out instead of the headline's random placement)")
    p.add_argument("--no-layouts", action="store_true", help="N > 1: skip the second (data-parallel) measurement of `layouts`")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL over xGMI (default); gloo lets several ranks share ONE GPU to exercise the multi-rank path")
    p.add_argument("--overlap-graph", action="store_true",
                   help="N > 1, RCCL: ALSO time the <layout>_rccl_overlap design captured into ONE HIP graph (the all-reduces on the "
                        "forked half-stream branches, the other half's GEMMs beside them: the graph form of north_star's overlap). "
                        "Opt-in and timed LAST: a capture of a collective that fails cannot always be recovered in-process "
                        "(tools/rccl_capture_probe.py is the one-GPU probe of the mechanism)")
    p.add_argument("--time-limit-s", type=float, default=0.0,
                   help="N > 1: wall-clock budget of the run; once it is spent the remaining OPTIONAL arms (rccl_overlap, dp, prefill, "
                        "engine) are skipped -- decided collectively, recorded in config.attempts -- so that rank 0 still prints its "
                        "line (the one-GPU pre-flight of the multi-rank paths sets it; 0 = no limit)")
    p.add_argument("--emulate-tp", type=int, default=0,
                   help="single GPU: run ONE rank's shard of a TP=k job with the collectives stubbed (tuning aid)")
    p.add_argument("--emulate-dp", type=int, default=0,
                   help="single GPU: run ONE replica's share of the batch of a DP=k job (tuning aid, with --emulate-tp)")
    return p.parse_args()


def live_pmc_traffic(B, ctx, nq, nkv, launches=4, timeout_s=240, probe_name="attn_pmc_probe.py", kernel_like="%paged_decode_kernel%",
                     probe_args=None):
    """roofline.traffic measured in this run: rocprofv3 --kernel-trace --pmc <counter> (one counter per pass) around
    tools/attn_pmc_probe.py, per-dispatch sums of the paged_decode kernel read back from the rocpd database. gfx950 correction of
    MI355X_MICROARCH.md's HBM section: FETCH_SIZE (KiB) reports half the bytes of a wide coalesced streaming read -> doubled;
    WRITE_SIZE (KiB) as reported. Returns (bytes per launch, provenance) or (None, None) when rocprofv3 is unavailable / fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    probe = os.path.join(ROOT, "tools", probe_name)
    pargs = [str(x) for x in (probe_args if probe_args is not None else (B, ctx, nq, nkv, launches))]
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            tmp = tempfile.mkdtemp(prefix="xm_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "--", sys.executable, probe] + pargs,
                               cwd="/tmp", env=env, timeout=timeout_s, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
                if not dbs:
                    return None, None
                cur = sqlite3.connect(dbs[0]).cursor()
                row = cur.execute("select sum(value), count(*) from counters_collection where counter_name = ? and "
                                  "kernel_name like ?", (counter, kernel_like)).fetchone()
                if not row or not row[1]:
                    return None, None
                got[counter] = float(row[0]) / float(row[1])          # KiB per dispatch
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        traffic = int(2.0 * got["FETCH_SIZE"] * 1024 + got["WRITE_SIZE"] * 1024)
        return traffic, (f"rocprofv3 --pmc in this run (tools/{probe_name}, {launches} launches per counter pass): FETCH_SIZE "
                         f"{got['FETCH_SIZE']:.0f} KiB x 2 (gfx950 correction) + WRITE_SIZE {got['WRITE_SIZE']:.0f} KiB per dispatch")
    except Exception as e:  # noqa: BLE001
        print(f"[bench] live PMC pass failed ({e!r}); quoting the committed record instead", file=sys.stderr)
        return None, None


def live_mfma_busy(kind, timeout_s=240):
    """north_star's "MFMA util on the quant GEMM", measured in this run: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one
    pass) over a few launches of the gate_up GEMM at M = 8192 in a child process (tools/gemm_one.py for int8 with per-row-quantised
    Gaussian operands, tools/fp8_gemm_one.py for fp8). busy = MFMA-busy cycles / (active cycles per XCD x 1024 SIMDs), the granted
    clock = active cycles per XCD / kernel duration. Returns a dict or None."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tool = os.path.join(ROOT, "tools", "gemm_one.py" if kind == "int8" else "fp8_gemm_one.py")
    tmp = tempfile.mkdtemp(prefix="xm_mfma_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", GEMM_DIST="gauss", GEMM_LAUNCHES="6")
        subprocess.run([exe, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "-d", tmp, "--", sys.executable,
                        tool, "8192", "37888", "3584"], cwd="/tmp", env=env, timeout=timeout_s, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        cur = sqlite3.connect(dbs[0]).cursor()
        vals = {}
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            row = cur.execute("select sum(value), count(*) from counters_collection where counter_name = ? and kernel_name like "
                              "'%gemm_p8%'", (c,)).fetchone()
            if not row or not row[1]:
                return None
            vals[c] = float(row[0]) / float(row[1])
        dur = cur.execute("select avg(end - start) from kernels where name like '%gemm_p8%'").fetchone()[0]
        cycles = vals["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        return {"mfma_busy": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 4),
                "clock_ghz": round(cycles / float(dur), 3) if dur else None, "us_under_pmc": round(float(dur) / 1e3, 1) if dur else None}
    except Exception as e:  # noqa: BLE001
        print(f"[bench] live MFMA-busy pass ({kind}) failed ({e!r}); quoting the committed record instead", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build_metadata(B, ctx, block_size, device, seed, placement="random"):
    """BatchInputBuilder-shaped decode metadata (framework/batch/batch_input_builder.cpp:739-830, 904-938).
    placement of a sequence's pages in the pool: "random" = a random permutation of the pool (a long-running server whose free
    list has been recycled; the headline, and the pessimistic case for the KV walk); "allocator" = what the reference's
    BlockManagerImpl hands a batch of whole prompts from a FRESH pool (framework/block/block_manager_impl.cpp:56-61: the free
    list is filled with descending ids and popped from the back, block 0 is the padding block; :66-80 allocate(n) pops n ids in a
    row): sequence i owns the ascending ids 1 + i * pages ... (i + 1) * pages."""
    pages = (ctx + block_size - 1) // block_size
    n_blocks = int(B * pages * 1.1) + 1
    g = torch.Generator().manual_seed(seed)
    if placement == "allocator":
        perm = (1 + torch.arange(B * pages, dtype=torch.int32)).view(B, pages)
    else:
        perm = torch.randperm(n_blocks, generator=g)[: B * pages].to(torch.int32).view(B, pages)
    # the product's own host-side builder (C++ behind the C ABI): one new token per sequence, ctx - 1 already cached
    from xllm_amd import attention
    bi = attention.build_batch_input([ctx - 1] * B, [ctx] * B, perm.tolist(), block_size)
    md = attention.build_attention_metadata(bi, is_prefill=False, is_chunked_prefill=False, device=device)
    return md, n_blocks


def cpu_baseline(args_model, mode, ctx, block_size):
    """The oracle (CPU restatement of the reference path) timed on this host's cores on a bounded sample:
    ONE decoder layer, decode step, B_s sequences at the full ctx; extrapolated to L layers + lm_head."""
    import ctypes as C
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    a = args_model
    Bs = 64  # bounded sample: ~10-30 s of CPU work on the host cores
    g = torch.Generator().manual_seed(0)
    pages = (ctx + block_size - 1) // block_size
    nb = Bs * pages
    kc = torch.randn(nb, block_size, a.n_kv_heads, a.head_dim, generator=g).bfloat16()
    vc = torch.randn(nb, block_size, a.n_kv_heads, a.head_dim, generator=g).bfloat16()
    table = torch.randperm(nb, generator=g).to(torch.int32).view(Bs, pages)
    kv_lens = torch.full((Bs,), ctx, dtype=torch.int32)
    cu_q = torch.arange(Bs + 1, dtype=torch.int32)
    H, I = a.hidden_size, a.intermediate_size
    qsz, kvsz = a.n_heads * a.head_dim, a.n_kv_heads * a.head_dim
    mk = lambda n, k: (torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int8), torch.rand(n, generator=g) * 1e-3)
    w_qkv, w_o, w_gu, w_dn = mk(qsz + 2 * kvsz, H), mk(H, qsz), mk(2 * I, H), mk(H, I)
    nw = (torch.rand(H, generator=g) + 0.5).bfloat16()
    x = torch.randn(Bs, H, generator=g).bfloat16()
    res = torch.randn(Bs, H, generator=g).bfloat16()
    cache = orc.build_cos_sin_cache(ctx, a.head_dim, a.rope_theta, torch.bfloat16)
    pos = torch.full((Bs,), ctx - 1)
    slots = (table[:, (ctx - 1) // block_size] * block_size + (ctx - 1) % block_size).to(torch.int32)

    def lin(inp, w):
        q, s = orc.scaled_quantize(inp)
        return orc.scaled_matmul(q, w[0], s, w[1], torch.bfloat16)

    def layer():
        h, r = x.clone(), res.clone()
        orc.fused_add_rms_norm(h, r, nw, a.rms_norm_eps)
        qkv = lin(h, w_qkv)
        q, k, v = qkv[:, :qsz], qkv[:, qsz:qsz + kvsz], qkv[:, qsz + kvsz:]
        orc.rotary_embedding(pos, q, k, cache, a.head_dim)
        orc.reshape_paged_cache(slots, k.unflatten(-1, (a.n_kv_heads, a.head_dim)),
                                v.unflatten(-1, (a.n_kv_heads, a.head_dim)), kc, vc)
        at = orc.paged_attention(q.unflatten(-1, (a.n_heads, a.head_dim)), kc, vc, cu_q, kv_lens, table,
                                 a.head_dim ** -0.5)
        o = lin(at, w_o)
        orc.fused_add_rms_norm(o, r, nw, a.rms_norm_eps)
        gu = lin(o, w_gu)
        act = torch.empty(Bs, I, dtype=torch.bfloat16)
        orc.act_and_mul(act, gu, "silu")
        return lin(act, w_dn)

    layer()  # warm (page-in)
    t0 = time.perf_counter()
    layer()
    t_layer = time.perf_counter() - t0
    # lm_head on the sample: [Bs,H] x [V,H]^T in 16-bit; time a 1/16 slice of V and scale
    Vs = a.vocab_size // 16
    lw = (torch.randn(Vs, H, generator=g) * 0.02).bfloat16()
    t0 = time.perf_counter()
    orc.matmul(x, lw)
    t_head = (time.perf_counter() - t0) * 16
    t_step = t_layer * a.n_layers + t_head
    return {"value": round(Bs / t_step, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle (OpenMP, {cores} threads): 1 decoder layer x 1 decode step over {Bs} of the batch's 256 sequences "
                      f"at ctx={ctx}, extrapolated x{a.n_layers} in layers; lm_head on {Bs} rows x 1/16 of the vocab, extrapolated "
                      f"x16 in columns; value = {Bs} sequences / that time, i.e. the full batch of 256 is assumed to run at the "
                      f"sample's rate (an extrapolation x{256 // Bs} in batch); t_layer={t_layer:.3f}s t_lm_head={t_head:.3f}s"}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU, and relay its
    output (the reference forks one worker per device the same way, runtime/worker_server.cpp:285-326). Never reports a
    one-GPU number for an N-GPU request."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if a.backend == "nccl" and ndev < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline_cfg1():
    """BASELINE.json configs[0]: Qwen2-0.5B fp32, bs = 1, ctx = 128, greedy decode "on the reference CPU path". The reference
    has no CPU backend, so the path is the end-to-end oracle model (oracle/model.py: the restated operators in the
    reference's order, pinned on the HuggingFace Qwen2 implementation) run in fp32 on this host's cores: prefill of 128
    tokens, then 16 greedy decode steps, timed whole (no sampling of layers, no extrapolation)."""
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import layers
    # bs = 1: every operator is a matrix-VECTOR product of at most 4864 x 896; 16 threads is where the oracle's OpenMP loops
    # stop scaling on such sizes (with 256 threads a decode step takes 35 s instead of 0.1 s: fork / join and spinning)
    cores = min(os.cpu_count() or 1, 16)
    orc.lib().orc_set_num_threads(cores)
    args = layers.ModelArgs.qwen2_0_5b()
    w = omodel.export_weights(layers.Qwen2Model(args, "16bit", torch.float32, "cpu", seed=1))
    om = omodel.OracleQwen2(args, w, torch.float32)
    prompt = torch.randint(0, args.vocab_size, (128,), generator=torch.Generator().manual_seed(0))
    t0 = time.perf_counter()
    omodel.greedy_generate(om, prompt, 1, 128, collect_logits=False)          # prefill + first token
    t_prefill = time.perf_counter() - t0
    if t_prefill > 40.0:   # a starved host: do not spend minutes of the bench run on the reported baseline
        orc.lib().orc_set_num_threads(os.cpu_count() or 1)
        return {"value": None, "unit": "tokens/s", "cores": cores, "kind": "port",
                "sample": f"oracle model, Qwen2-0.5B fp32 bs=1: prefill of 128 tokens took {t_prefill:.1f}s on this host; "
                          f"decode leg skipped", "prefill_tokens_per_s": round(128 / t_prefill, 2)}
    n_new = 17
    t0 = time.perf_counter()
    omodel.greedy_generate(om, prompt, n_new, 128, collect_logits=False)
    t_all = time.perf_counter() - t0
    t_decode = max(t_all - t_prefill, 1e-9)
    orc.lib().orc_set_num_threads(os.cpu_count() or 1)
    return {"value": round((n_new - 1) / t_decode, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle model (OpenMP, {cores} threads), Qwen2-0.5B fp32 bs=1: prefill of 128 tokens {t_prefill:.2f}s, "
                      f"{n_new - 1} greedy decode steps {t_decode:.2f}s; whole model, nothing extrapolated",
            "prefill_tokens_per_s": round(128 / t_prefill, 2)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % ndev  # gloo self-test: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from xllm_amd import layers, parallel
    from xllm_amd.attention import KVCache

    if a.config.endswith("-slice"):
        if world != 1:
            raise SystemExit(f"--config {a.config} is a one-GPU slice of a TP=8 job: run it with --gpus 1")
        import bench_slices
        print(json.dumps(bench_slices.run(a, dev)), flush=True)
        return
    model_name, mode, gbatch, ctx = CONFIGS[a.config]
    margs = getattr(layers.ModelArgs, model_name)()
    block_size = 128
    dtype = torch.bfloat16
    from xllm_amd import ops

    def pick_layout():
        """how N GPUs share the fixed global batch. auto = what `north_star` names: tensor parallel over RCCL / xGMI (heads / columns
        per rank, SUM all-reduce after o_proj and down_proj, logits all-gather); 28 heads divide over 1, 2, 4, 7 ... ranks, so N = 8
        is TP = 4 inside two replicas (qwen2_attention.cpp:54-65). The data-parallel line is measured beside it (`layouts`)."""
        if world == 1:
            return "single", 1
        if a.layout == "dp":
            return "dp", 1
        if a.layout == "tp4dp2" or (a.layout == "auto" and world == 8):
            if world % 4:
                raise SystemExit("--layout tp4dp2 needs a multiple of 4 GPUs")
            return "tp4dp2", 4
        if a.layout == "tp" or margs.n_heads % world == 0:
            if margs.n_heads % world:
                raise SystemExit(f"--layout tp: {margs.n_heads} heads do not divide over {world} ranks "
                                 f"(qwen2_attention.cpp:54); use dp or tp4dp2")
            return "tp", world
        return "dp", 1

    layout, tp_size = pick_layout()

    # per-launch HIP events around the dominant kernel (paged decode attention) on the launch stream
    attn_events = []
    orig_paged = ops.paged_attention
    record = {"on": False}

    def timed_paged(*args, **kw):
        if not record["on"]:
            return orig_paged(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_paged(*args, **kw)
        e1.record()
        attn_events.append((e0, e1))
        return out

    ops.paged_attention = timed_paged
    orig_fused = ops.paged_decode_attention_int8

    def timed_fused(*args, **kw):  # the N1-fused decode attention (same kernel, int8 epilogue)
        if not record["on"]:
            return orig_fused(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_fused(*args, **kw)
        e1.record()
        if out is not None:
            attn_events.append((e0, e1))
        return out

    ops.paged_decode_attention_int8 = timed_fused

    def sync_all():
        torch.cuda.synchronize()        # this rank's work is done ...
        if world > 1:
            dist.barrier()              # ... and so is everybody's
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def build(tp_sz):
        """model shard + KV caches + metadata of this rank for TP = tp_sz inside world / tp_sz replicas"""
        dp_sz = world // tp_sz
        if gbatch % dp_sz:
            raise SystemExit(f"global batch {gbatch} does not divide over {dp_sz} replicas")
        tp_pg, dp_rank = (parallel.make_tp_dp_groups(world, rank, tp_sz) if world > 1 else (None, 0))
        if tp_pg is not None and tp_sz > 1 and not a.no_oneshot_allreduce:
            # the one-shot xGMI all-reduce for the per-layer sums (<= 8 MiB), self-tested at set-up; RCCL when the test fails
            tp_pg.enable_oneshot(dev, 8 << 20)
            if tp_pg.oneshot is not None:
                tp_pg.oneshot.timeout_s = 10.0   # bounded waits stay bounded; first launches (code load) may be slow on one rank
            if rank == 0 and tp_pg.oneshot is None:
                print(f"[bench] one-shot all-reduce not used: {tp_pg.oneshot_note}", file=sys.stderr)
        B = gbatch // dp_sz
        if a.emulate_dp > 1 and world == 1:
            B = gbatch // a.emulate_dp
        if a.emulate_tp > 1 and world == 1:
            class _StubPG(parallel.ProcessGroup):  # shard shapes of TP=k, no exchange: per-rank compute only
                exchange_stubbed = True   # row-parallel GEMM -> ONE slab consumer (add + norm + quant), as the one-shot kernel does

                def allreduce(self, x):
                    return None

                def allgather(self, x):
                    return x.unsqueeze(0).expand(self._world, *x.shape)
            tp_pg = _StubPG(None, 0, a.emulate_tp)
            tp_sz = a.emulate_tp
        model = layers.Qwen2Model(margs, mode, dtype, dev, seed=1234, tp=tp_pg, fuse=not a.no_fuse)
        md, n_blocks = build_metadata(B, ctx, block_size, dev, seed=dp_rank)
        nkv_l = model.layers[0].nkv
        gen = torch.Generator(device=dev).manual_seed(99 + rank)
        kv_caches = []
        for _ in model.layers:
            kc = torch.empty(n_blocks, block_size, nkv_l, margs.head_dim, dtype=dtype, device=dev).normal_(generator=gen)
            vc = torch.empty(n_blocks, block_size, nkv_l, margs.head_dim, dtype=dtype, device=dev).normal_(generator=gen)
            kv_caches.append(KVCache(kc, vc))
        tokens = torch.randint(0, margs.vocab_size, (B,), device=dev, generator=gen)
        positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
        return dict(model=model, md=md, n_blocks=n_blocks, kv_caches=kv_caches, tokens=tokens, positions=positions, B=B,
                    tp_pg=tp_pg, tp_size=tp_sz, dp_size=dp_sz, nkv_l=nkv_l)

    def time_decode(w, steps, overlap_arm=False, overlap_graph=False):
        """warm up, capture (one graph without collectives, piecewise graphs around eager collectives, or -- one-shot kernel --
        one graph WITH them), time `steps` replays; returns the timing record and the eager step function.
        overlap_arm: tensor parallel with the per-layer all-reduces on RCCL's own stream under the dual micro-batch executor
        (north_star's "RCCL all-reduce overlapped with the next GEMM on a side HIP stream": the next GEMM is the OTHER half's,
        parallel_state_async.cpp:72-84 / enable_multi_stream_parallel), launched eagerly (collectives from two streams do not go
        through the piecewise capture)"""
        model, md, kv_caches, tokens, positions, B = w["model"], w["md"], w["kv_caches"], w["tokens"], w["positions"], w["B"]
        tp_sz, tp_pg = w["tp_size"], w["tp_pg"]
        dual = layers.DualBatchDecoder(model, md, B) if ((a.dual and world == 1 or overlap_arm) and not a.no_fuse and mode == "int8") else None

        def step():
            hidden = dual.forward(tokens, positions, kv_caches) if dual is not None else model.forward(tokens, positions, md, kv_caches)
            return model.greedy_tokens(hidden)     # lm_head + Sampler::greedy_sample in one pass (round 4); [B, V] logits never written

        sync_all()     # ranks build their shards at different speeds: nobody enters the first collective seconds before a peer
        for _ in range(a.warmup):
            step()
        sync_all()
        # decode runs under HIP-graph replay in the reference (runtime/dcu_graph_executor_impl.h): capture one step
        # (every op of the C ABI is capture-safe: no host sync, no allocation inside) and replay it.
        graph = None
        # no collective inside the step (one GPU, or data-parallel replicas): ONE graph; tensor parallel: piecewise graphs
        # overlap_graph (opt-in, --overlap-graph): the dual micro-batch step with its RCCL all-reduces captured into ONE graph -- under
        # capture launch_reduce issues the collective in place on the half's forked stream (parallel.launch_reduce), so the graph holds
        # it as a node of that branch and the other half's GEMMs as nodes of the other branch
        use_graph = (not a.no_graph) and (not overlap_arm or overlap_graph) and \
                    (tp_sz == 1 or ((a.graph or overlap_graph) and a.backend == "nccl"))
        piecewise = (not a.no_graph) and not overlap_arm and tp_sz > 1 and not use_graph
        if piecewise:
            # TP > 1: one graph per run of kernels between two EAGER collectives (xllm_amd/parallel.py::PiecewiseGraph); RCCL / gloo
            # never run inside a capture. With the one-shot kernel the per-layer sums are kernels INSIDE the pieces: the only eager
            # collective left is the logits all-gather.
            try:
                step()
                pw = parallel.PiecewiseGraph()
                pw.capture(step)
                graph = pw
                for _ in range(2):
                    graph.replay()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] piecewise graph capture failed ({e!r}); falling back to eager launches", file=sys.stderr)
                graph = None
        if use_graph:
            try:
                cap_stream = torch.cuda.Stream()
                cap_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap_stream):
                    step()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=cap_stream):
                        step()
                torch.cuda.current_stream().wait_stream(cap_stream)
                graph = g
                for _ in range(2):
                    graph.replay()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] graph capture failed ({e!r}); falling back to eager launches", file=sys.stderr)
                graph = None
        run_step = (lambda: graph.replay()) if graph is not None else step
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if tp_pg is not None:
            tp_pg.check_agreed()   # the one-shot kernel's bounded waits: a timed-out launch on ANY rank voids the measurement and
                                   # raises on EVERY rank (collective), so all ranks take the RCCL re-timing branch together
        ms = elapsed / steps * 1e3
        exposed = 0.0 if tp_sz == 1 else None
        eager_collectives = 0
        if piecewise and graph is not None:
            eager_collectives = sum(1 for it in graph.items if not isinstance(it, torch.cuda.CUDAGraph))
            # the same piecewise replay with the eager collectives left out (results are then wrong, timing only): the difference
            # is the communication time that is NOT hidden under compute. (All-reduces that run as one-shot kernels inside the
            # pieces stay in: their cost is part of both figures.)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay(skip_collectives=True)
            sync_all()
            t_nc = max_over_ranks(time.perf_counter() - t0)
            exposed = round(max(ms - t_nc / steps * 1e3, 0.0), 4)
        return dict(ms_per_step=ms, tok_s=gbatch * steps / elapsed, graph=graph, piecewise=piecewise, exposed_comm_ms=exposed,
                    eager_collectives=eager_collectives, dual=dual, step=step)

    w = build(tp_size)
    model, md, n_blocks, kv_caches, tokens, positions, B = (w["model"], w["md"], w["n_blocks"], w["kv_caches"], w["tokens"],
                                                             w["positions"], w["B"])
    tp_pg, tp_size, dp_size, nkv_l = w["tp_pg"], w["tp_size"], w["dp_size"], w["nkv_l"]
    # ---- the headline timing, made hard to lose (round-4 review, next #5): a multi-GPU lease is rare, so
    #   * every attempt is guarded and its verdict AGREED over the ranks (a failure on any rank counts for all, so every rank takes
    #     the same next branch and issues the same collectives);
    #   * tensor parallel: the conservative arm `north_star` names -- the group's own all-reduce (RCCL over xGMI) in stream between
    #     piecewise graphs -- is timed FIRST; only then the one-shot kernel (self-tested at set-up) gets its run. The headline is the
    #     one-shot run when it came out clean on every rank (bounded waits, status word checked after the timed region), the RCCL
    #     run otherwise; both are reported in `layouts`;
    #   * an attempt whose graph capture fails is repeated with eager launches before anything is given up;
    #   * if nothing could be timed, rank 0 still prints ONE valid JSON line (value null, the errors in `config.attempts`).
    attempts = []

    def attempt(label, fn):
        res, err = None, None
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            err = repr(e)
            print(f"[bench] rank {rank}: attempt '{label}' failed: {err}", file=sys.stderr)
        failed = err is not None
        if world > 1:
            flag = torch.tensor([1.0 if failed else 0.0], device=dev if a.backend == "nccl" else "cpu")
            try:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                failed = float(flag.item()) > 0
            except Exception as e:  # noqa: BLE001
                failed, err = True, (err or "") + f" | agreement all-reduce failed: {e!r}"
        attempts.append({"arm": label, "ok": False, "error": err or "failed on another rank"} if failed else {"arm": label, "ok": True})
        return None if failed else res

    t_run0 = time.perf_counter()
    shared_gpu = world > 1 and world > torch.cuda.device_count()   # the gloo pre-flight: several ranks on ONE GPU (rank-invariant)

    def skip_arm(label):
        """collective: True when an OPTIONAL arm is to be left out -- the run's wall-clock budget (--time-limit-s) is spent on ANY
        rank (agreed: a rank that went on alone would sit in collectives its peers never issue). Recorded in config.attempts."""
        over = a.time_limit_s > 0 and (time.perf_counter() - t_run0) > a.time_limit_s
        if world > 1:
            flag = torch.tensor([1.0 if over else 0.0], device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            over = float(flag.item()) > 0
        if over:
            attempts.append({"arm": label, "ok": False, "skipped": f"--time-limit-s {a.time_limit_s:g} spent before this arm"})
        return over

    def timed(label, **kw):
        got = attempt(label, lambda: time_decode(w, a.steps, **kw))
        if got is None and not a.no_graph:
            a.no_graph = True            # eager launches: no capture to fail
            try:
                got = attempt(label + " (eager, no graph)", lambda: time_decode(w, a.steps, **kw))
            finally:
                a.no_graph = False
        return got

    rccl_ranks_seen = None
    if world > 1:      # every rank adds a one through the group the step will use: the line says how many ranks RCCL / gloo saw
        def _count():
            ones = torch.ones(1, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(ones)
            return int(ones.item())
        rccl_ranks_seen = attempt("world all-reduce of ones", _count)
    r, r_rccl, r_oneshot = None, None, None
    if world > 1 and w["tp_pg"] is not None and w["tp_size"] > 1:
        saved_oneshot = w["tp_pg"].oneshot
        w["tp_pg"].oneshot = None
        r_rccl = timed(f"{layout}_rccl")
        w["tp_pg"].oneshot = saved_oneshot
        if saved_oneshot is not None:
            def _oneshot_run():
                got = time_decode(w, a.steps)
                if os.environ.get("XLLM_MI355_BENCH_INJECT_ONESHOT_FAILURE") == "1":
                    raise RuntimeError("injected one-shot failure (test of the fallback)")
                return got
            r_oneshot = attempt(f"{layout}_oneshot", _oneshot_run)
            if r_oneshot is None:
                w["tp_pg"].oneshot_note = "dropped: its timed run failed on some rank (see config.attempts); RCCL is the headline"
                if rank == 0:
                    print(f"[bench] one-shot all-reduce {w['tp_pg'].oneshot_note}", file=sys.stderr)
                w["tp_pg"].oneshot = None
        r = r_oneshot if r_oneshot is not None else r_rccl
    else:
        r = timed("headline")
    if r is None:
        if rank == 0:
            print(json.dumps({"metric": "decode tokens/s, Qwen2-7B int8 bs=256 ctx=4096", "value": None, "unit": "tokens/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "int8" if mode == "int8" else "bf16",
                              "data": "synthetic", "config": {"workload": f"{model_name} decode step", "layout": layout,
                                                              "attempts": attempts, "rccl_ranks_seen": rccl_ranks_seen},
                              "error": "no arm of the decode step could be timed"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(1)
    ms_per_step, tok_s, graph, piecewise, exposed_comm_ms, dual, step = (r["ms_per_step"], r["tok_s"], r["graph"], r["piecewise"],
                                                                        r["exposed_comm_ms"], r["dual"], r["step"])
    # timed right after the headline (same clocks, same temperature): the page placement is the only difference
    alloc_pages = None
    if world == 1 and tp_size == 1 and dual is None and not a.no_allocator_pages and a.emulate_tp <= 1 and a.emulate_dp <= 1:
        def _alloc_pages():
            w2 = dict(w)
            w2["md"], nb2 = build_metadata(B, ctx, block_size, dev, seed=0, placement="allocator")
            assert nb2 == n_blocks
            r2 = time_decode(w2, a.steps)
            return {"ms_per_step": round(r2["ms_per_step"], 4), "tokens_per_s": round(r2["tok_s"], 2),
                    "what": "the SAME step and KV pool with the block table of a fresh BlockManagerImpl pool (block_manager_impl.cpp:56-80: "
                            "each sequence's pages are consecutive ascending ids) instead of the headline's random placement; not the headline"}
        alloc_pages = attempt("allocator-order pages", _alloc_pages)

    # exchange accounting (reference: 2 all-reduces per layer + the logits all-gather, linear.cpp:1518-1520, 712-714)
    collectives_per_step = (2 * len(model.layers) + 1) if tp_size > 1 else 0
    allreduce_kind = tp_pg.allreduce_kind() if (tp_pg is not None and tp_size > 1 and world > 1) else None

    # roofline leg: per-launch HIP events around the dominant kernel (paged decode attention) on the launch
    # stream, over eager steps of the same workload (events cannot be read back from inside a replayed graph)
    record["on"] = True
    for _ in range(min(max(a.steps, 1), 20)):     # up to 20 eager steps = 560 launches (round-5 review: 3 steps were 84)
        step()
    sync_all()
    record["on"] = False

    attn_ms = sum(e0.elapsed_time(e1) for e0, e1 in attn_events) / max(len(attn_events), 1)

    def attention_path_in_graph():
        """the attention launches of the step by themselves -- every layer's call on its own KV cache, exactly what the layer issues
        (the fused int8 epilogue, or attention + split-KV merge + scaled_quantize when the plan declines it) -- captured into ONE
        graph and replayed: microseconds per layer without the host's launch gaps. For the 336-us launches of the headline the
        eager HIP events above agree with it; for the short launches of the per-rank shapes (50-60 us) the eager figure is an
        upper bound (the host cannot enqueue fast enough) and this is the kernel-side number."""
        L0 = model.layers[0]
        qa = torch.randn(B, L0.nq * L0.d, device=dev, dtype=dtype)

        def run():
            for layer, kvc in zip(model.layers, kv_caches):
                if mode == "int8" and not a.no_fuse:
                    layer.attention_kernel(qa, md, kvc)
                else:
                    ops.paged_attention(qa.unflatten(-1, (L0.nq, L0.d)), kvc.k_cache, kvc.v_cache, None, md.kv_seq_lens, md.block_table,
                                        1, md.max_seq_len, L0.attn.scale, False, L0.attn.window_left)
        try:
            record["on"] = False
            run()
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                run()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    run()
                for _ in range(2):
                    gr.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    gr.replay()
                e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (20 * len(model.layers))
        except Exception as e:  # noqa: BLE001
            print(f"[bench] attention-in-graph timing failed: {e!r}", file=sys.stderr)
            return None

    attn_graph_ms = attention_path_in_graph() if world == 1 else None
    nq_l = model.layers[0].nq
    d = margs.head_dim
    # algorithmic bytes per launch: K+V of every cached token once + Q in + O out
    attn_bytes = B * (ctx * nkv_l * d * 2 * 2 + 2 * nq_l * d * 2)
    # Launches of ~50 us (cfg2, the per-rank shapes): HIP events around an EAGER launch also time the host's gap between the event
    # record and the kernel's enqueue -- 55 us on a quiet host, 167 us on a busy one for the same 50-us kernel (round 6). When the
    # eager reading is > 15 % above the same launches replayed from one graph (HIP events around the replays, on the launch
    # stream, no host in between), the graph reading is the launch duration; both are reported.
    eager_ms = attn_ms
    timing = "HIP events around each eager launch"
    if attn_graph_ms is not None and attn_graph_ms > 0 and attn_ms > 1.15 * attn_graph_ms:
        attn_ms = attn_graph_ms
        timing = ("HIP events around the launches replayed from one HIP graph (the eager-launch events read %.4f ms: host launch gaps)"
                  % eager_ms)
    achieved = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0
    traffic, traffic_source = None, None
    if world == 1 and tp_size == 1 and a.config in ("cfg3", "cfg2") and not a.no_pmc:
        # HBM bytes per launch of the dominant kernel from the PMC counters, collected IN THIS RUN (round-3 review, weak #12): two
        # rocprofv3 passes (one counter each, as MI355X_MICROARCH.md prescribes) over tools/attn_pmc_probe.py = the same kernel at
        # the same shape in a child process
        traffic, traffic_source = live_pmc_traffic(B, ctx, nq_l, nkv_l)
    pmc = os.path.join(ROOT, "profiles", "decode_attn_pmc.json")
    if traffic is None and os.path.exists(pmc) and world == 1 and tp_size == 1 and a.config == "cfg3":
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            traffic_source = "profiles/decode_attn_pmc.json (rocprofv3 --pmc passes of an earlier run, not this run)"
        except Exception:
            traffic = None

    # the other way N GPUs can share the batch, measured in the same run (a legitimate deployment answer, reported, not
    # substituted): N data-parallel replicas, no data-path exchange
    layouts = None
    if world > 1 and tp_size > 1 and a.layout == "auto" and not a.no_layouts:
        layouts = {layout: {"ms_per_step": round(ms_per_step, 4), "tokens_per_s": round(tok_s, 2),
                            "collectives_per_step": collectives_per_step, "allreduce": allreduce_kind,
                            "exposed_comm_ms": exposed_comm_ms}}
        # the two other exchange designs of the SAME tensor-parallel layout, from the same lease (round-3 review, next #8): the
        # group's own all-reduce (RCCL) in stream between piecewise graphs, and RCCL on its own stream under the dual micro-batch
        # executor (the overlap north_star describes). Each arm is optional: a failure is recorded, never fatal. Every rank runs
        # the same code, so the collectives of an arm are issued in the same order everywhere.
        saved_oneshot = tp_pg.oneshot if tp_pg is not None else None
        fmt = lambda ra, overlap: {"ms_per_step": round(ra["ms_per_step"], 4), "tokens_per_s": round(ra["tok_s"], 2),
                                   "collectives_per_step": collectives_per_step, "allreduce": "rccl" if a.backend == "nccl" else a.backend,
                                   "exposed_comm_ms": ra["exposed_comm_ms"], "micro_batches": 2 if overlap else 1,
                                   "hip_graph": ("piecewise" if ra["piecewise"] else bool(ra["graph"])) if ra["graph"] is not None else False}
        # <layout>_rccl was timed FIRST (above); when the one-shot run did not make the headline, the headline IS that run
        layouts[layout + "_rccl"] = ({"same_as": layout} if r is r_rccl else fmt(r_rccl, False)) if r_rccl is not None else \
            {"error": "see config.attempts"}
        arm = layout + "_rccl_overlap"
        try:
            tp_pg.oneshot = None
            ra = None if skip_arm(arm) else attempt(arm, lambda: time_decode(w, a.steps, overlap_arm=True))
            if ra is None:
                layouts[arm] = {"error": "see config.attempts"}
            else:
                layouts[arm] = fmt(ra, True)
                if ra.get("dual") is not None:
                    ra["dual"].close()
            del ra
        finally:
            tp_pg.oneshot = saved_oneshot
        torch.cuda.empty_cache()
        def _dp():
            w2 = build(1)
            r2 = time_decode(w2, a.steps)
            return {"ms_per_step": round(r2["ms_per_step"], 4), "tokens_per_s": round(r2["tok_s"], 2), "collectives_per_step": 0,
                    "allreduce": None, "exposed_comm_ms": 0.0, "per_gpu_batch": w2["B"]}
        if shared_gpu:
            # N full replicas of the model next to the TP shards do not fit ONE GPU (8 x 33 GB at N = 8: round 5's pre-flight ran
            # out of memory here and never printed its line); the arm needs one GPU per rank
            attempts.append({"arm": "dp", "ok": False, "skipped": "ranks share one GPU: N replicas of the whole model do not fit it"})
            layouts["dp"] = {"skipped": "ranks share one GPU"}
        elif skip_arm("dp"):
            layouts["dp"] = {"error": "see config.attempts"}
        else:
            layouts["dp"] = attempt("dp", _dp) or {"error": "see config.attempts"}
        torch.cuda.empty_cache()
        if a.overlap_graph and a.backend == "nccl":
            arm = layout + "_rccl_overlap_graph"
            try:
                tp_pg.oneshot = None
                ra = None if skip_arm(arm) else attempt(arm, lambda: time_decode(w, a.steps, overlap_arm=True, overlap_graph=True))
                layouts[arm] = fmt(ra, True) if ra is not None else {"error": "see config.attempts"}
                if ra is not None and ra.get("dual") is not None:
                    ra["dual"].close()
                del ra
            finally:
                tp_pg.oneshot = saved_oneshot
            torch.cuda.empty_cache()

    prefill = None
    if not a.no_prefill and a.config == "cfg3" and not (world > 1 and skip_arm("prefill")):
        prefill = prefill_leg(model, margs, kv_caches, block_size, ctx, dev, world, tp_size, dp_size, sync_all)

    gemm_info = None
    if world == 1 and tp_size == 1 and a.config == "cfg3" and not a.no_gemm:
        gemm_info = gemm_leg(dev, live_pmc=not a.no_pmc)

    per_rank = None
    if world == 1 and tp_size == 1 and a.config == "cfg3" and not a.no_per_rank and a.emulate_tp <= 1 and a.emulate_dp <= 1:
        per_rank = per_rank_emulated()

    shim_info = None
    if a.via_shim and world == 1 and tp_size == 1 and mode == "int8":
        shim_info = via_shim_leg(model, margs, md, kv_caches, tokens, positions, a.steps, a.warmup)

    engine_info = None
    if world == 1 and not a.no_engine and dual is None and tp_size == 1:
        engine_info = engine_leg(model, kv_caches, B, ctx, block_size, n_blocks, dev, a.steps, margs)

    if a.micro and rank == 0:
        micro(model, md, kv_caches, tokens, positions, B, sys.stderr)

    if rank == 0:
        out = {
            "metric": "decode tokens/s, Qwen2-7B int8 bs=256 ctx=4096" if a.config == "cfg3" else f"decode tokens/s ({a.config})",
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int8" if mode == "int8" else "bf16", "data": "synthetic",
            "config": {"workload": f"{model_name} {'W8A8 int8' if mode == 'int8' else 'bf16'} decode step, "
                                   f"global_batch={gbatch} ctx={ctx}, paged KV block={block_size} bf16, "
                                   f"{margs.n_layers} layers + lm_head + argmax, random-init weights",
                       "global_batch": gbatch, "ctx": ctx, "per_gpu_batch": B,
                       "block_table": "random placement of every sequence's pages in a pool of 1.1 x the pages in use",
                       "parallelism": (f"dp{dp_size}" if tp_size == 1 and dp_size > 1 else
                                       f"tp{tp_size}" + (f"xdp{dp_size}" if dp_size > 1 else "")),
                       "layout": layout if world > 1 else "single", "collectives_per_step": collectives_per_step,
                       "allreduce": allreduce_kind,
                       "allreduce_note": (tp_pg.oneshot_note if (tp_pg is not None and tp_size > 1 and world > 1) else None),
                       "allreduce_grid_limit": (getattr(getattr(tp_pg, "oneshot", None), "grid_limit", None)
                                                if (tp_pg is not None and tp_size > 1 and world > 1) else None),
                       "eager_collectives_per_step": r["eager_collectives"] if tp_size > 1 else 0,
                       "rccl_ranks_seen": rccl_ranks_seen, "attempts": attempts,
                       "exposed_comm_ms": exposed_comm_ms,
                       "quant_fusion": not a.no_fuse, "micro_batches": 2 if dual is not None else 1,
                       "hip_graph": ("piecewise" if piecewise else True) if graph is not None else False},
            "roofline": {"bound": "hbm", "kernel": "paged_decode_kernel (+ split-KV merge when the launch splits)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": attn_bytes, "avg_launch_ms": round(attn_ms, 4), "timing": timing,
                         "eager_events_ms": round(eager_ms, 4), "launches_timed": len(attn_events),
                         "attention_path_ms_in_graph": None if attn_graph_ms is None else round(attn_graph_ms, 4),
                         "attention_path_frac_in_graph": None if not attn_graph_ms else round(attn_bytes / (attn_graph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if layouts is not None:
            out["layouts"] = layouts
        if prefill is not None:
            out["prefill"] = prefill
        if engine_info is not None:
            out["engine"] = engine_info
        if shim_info is not None:
            out["via_shim"] = shim_info
        if gemm_info is not None:
            out["gemm"] = gemm_info
        if per_rank is not None:
            out["per_rank_emulated"] = per_rank
        if alloc_pages is not None:
            out["allocator_order_pages"] = alloc_pages
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(margs, mode, ctx, block_size)
            if a.config == "cfg3":
                out["cpu_baseline"]["cfg1"] = cpu_baseline_cfg1()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()




def per_rank_emulated():
    """what ONE rank of the 8-GPU layouts computes per step, measured on this GPU with the shard shapes and NO exchange
    (children of this process: `bench.py --emulate-tp 4 --emulate-dp 2` = a rank of TP4 x DP2, B = 128, 7 q heads + 1 kv head,
    a quarter of every weight; `--emulate-dp 8` = a DP replica, B = 32, the whole model). The driver measures the real scaling;
    these bound it from above: N-GPU speed-up <= step(1 GPU) / step(rank)."""
    import subprocess
    out = {}
    for name, flags in (("tp4dp2_rank", ["--emulate-tp", "4", "--emulate-dp", "2"]), ("dp8_replica", ["--emulate-dp", "8"])):
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--no-cpu-baseline", "--no-prefill", "--no-gemm", "--no-engine",
                                                                      "--no-pmc", "--no-per-rank", "--steps", "20", "--warmup", "3"]
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            r = d["roofline"]
            out[name] = {"ms_per_step": d["ms_per_step"], "per_rank_batch": d["config"]["per_gpu_batch"],
                         "attention_us_eager_events": round(r.get("eager_events_ms", r["avg_launch_ms"]) * 1e3, 1),
                         "attention_frac_eager_events": round(r["bytes_per_launch"] / (r.get("eager_events_ms", r["avg_launch_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "attention_path_us_in_graph": None if r.get("attention_path_ms_in_graph") is None else round(r["attention_path_ms_in_graph"] * 1e3, 1),
                         "attention_path_frac_in_graph": r.get("attention_path_frac_in_graph"),
                         "attention_bytes_per_launch": r["bytes_per_launch"], "command": "bench.py " + " ".join(flags)}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


def gemm_leg(dev, live_pmc=True):
    """the quantised GEMM by itself (north_star: ">= 60 % fp8 MFMA util on quant GEMM at TP = 1"): Qwen2-7B gate_up
    (N = 37888, K = 3584) at the prefill M = 8192 and at a decode M = 128, int8 and fp8, timed with HIP events over a
    100-launch graph (weights rotate over 4 copies = 543 MB > Infinity Cache). `frac_of_peak` is achieved / 5 PFLOP/s (the
    dense 8-bit MFMA peak, MI355X_MICROARCH.md); at M = 128 the bound is the weight stream, so `frac_of_hbm` is given too.
    `mfma_busy` is a PMC figure (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE x SIMDs): since round 4 it is MEASURED IN THIS RUN by a
    rocprofv3 --pmc pass over the same GEMM in a child process (live_mfma_busy); the committed record is quoted only when that
    pass is unavailable."""
    from xllm_amd import ops
    N, K = 37888, 3584
    out = {"shape": f"gate_up N={N} K={K}", "peak_tops": 5000.0, "hbm_peak_gbs": HBM_PEAK_GBS,
           "mfma_busy": {"int8_M8192": 0.638, "fp8_M8192": 0.700, "int8_M8192_clock_ghz": 1.57,
                         "source": "profiles/r03_prefill.txt (int8, round 3) and profiles/r01_gemm_p8_pmc.txt (fp8): rocprofv3 --pmc "
                                   "passes of earlier runs, not measured in this run"}}
    if live_pmc:
        li, lf = live_mfma_busy("int8"), live_mfma_busy("fp8")
        if li is not None and lf is not None:
            out["mfma_busy"] = {"int8_M8192": li["mfma_busy"], "fp8_M8192": lf["mfma_busy"], "int8_M8192_clock_ghz": li["clock_ghz"],
                                "fp8_M8192_clock_ghz": lf["clock_ghz"], "int8_us_under_pmc": li["us_under_pmc"],
                                "fp8_us_under_pmc": lf["us_under_pmc"],
                                "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in this run (tools/gemm_one.py "
                                          "GEMM_DIST=gauss / tools/fp8_gemm_one.py, 8192 x 37888 x 3584): busy = MFMA-busy cycles / "
                                          "(active cycles per XCD x 1024 SIMDs)"}
    g = torch.Generator(device=dev).manual_seed(3)
    copies = 4
    for kind in ("int8", "fp8"):
        if kind == "int8":
            ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g) for _ in range(copies)]
            wps = [ops.pack_weight_i8(w) for w in ws]
        else:
            ws = [(torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn) for _ in range(copies)]
            wps = [ops.pack_weight_fp8(w) for w in ws]
        w_s = torch.rand(N, device=dev, generator=g) * 0.02 + 0.01
        for M in (8192, 256, 128):       # 256 = the headline's decode batch (round-4 review, weak #5: the shape the step runs was absent)
            if kind == "int8":
                a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
                a_s = torch.rand(M, device=dev, generator=g) * 0.01
                fn = lambda i: ops.scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies])
            else:
                a = (torch.randn(M, K, device=dev, generator=g) * 2).to(torch.float8_e4m3fn)
                a_s = torch.rand(M, device=dev, generator=g) * 0.05 + 0.01
                fn = lambda i: ops.fp8_scaled_matmul(a, ws[i % copies], a_s, w_s, torch.bfloat16, b_packed=wps[i % copies])
            n = 20 if M > 512 else 100
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                fn(0)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for i in range(n):
                        fn(i)
                reps = 2 if M > 512 else 10   # (short launches: >= 30 ms of replays on both sides, the clock governor needs it)
                for _ in range(1 if M > 512 else reps):
                    gr.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    gr.replay()
                e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * n)
            tops = 2.0 * M * N * K / us / 1e6
            out[f"{kind}_M{M}"] = {"us": round(us, 1), "tops": round(tops, 1), "frac_of_peak": round(tops / 5000.0, 4),
                                    "gbs": round((N * K + M * K + 2 * M * N) / us / 1e3, 1),
                                    "frac_of_hbm": round((N * K + M * K + 2 * M * N) / us / 1e3 / HBM_PEAK_GBS, 4)}
        del ws, wps
    torch.cuda.empty_cache()
    out["decode_layer_M256_nonattention"] = decode_layer_gemms(dev, 256)
    return out


def decode_layer_gemms(dev, M):
    """the NON-ATTENTION part of ONE Qwen2-7B W8A8 decoder layer at the headline's decode batch, as the launch pairs the fused step
    issues (xllm_amd/layers.py): qkv GEMM -> slabs -> dequant + RoPE + KV write; o_proj GEMM -> slabs -> add + RMSNorm + int8 quant;
    gate_up GEMM with SiLU.mul in its epilogue -> quantising pass; down_proj GEMM -> slabs -> add + RMSNorm + quant. Each pair is
    timed by itself over a graph of 50 repetitions on rotating weight copies (6 x the weights > Infinity Cache); `floor_us` =
    max(weight stream at 8 TB/s, int8 MFMA at 5 POP/s) of the GEMM. `sum_us` is what the review's per-layer budget refers to
    (round 4: 112.8 us of GEMMs + 29 us of row-wise kernels in the kernel trace)."""
    from xllm_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    H, I, nq, nkv, d = 3584, 18944, 28, 4, 128
    QKV = (nq + 2 * nkv) * d
    copies = 6
    out, total = {}, 0.0
    a_scale = torch.rand(M, device=dev, generator=g) * 0.01 + 0.001
    bs, nb = 128, M + 8
    kc = torch.zeros(nb, bs, nkv, d, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    slots = (torch.arange(M, device=dev, dtype=torch.int32) * bs + 5)
    pos = torch.full((M,), 4095, dtype=torch.int64, device=dev)
    from xllm_amd import layers
    cos_sin = layers.build_cos_sin_cache(layers.ModelArgs.qwen2_7b(), torch.bfloat16, dev, 8192)
    norm_w = (torch.rand(H, device=dev, generator=g) + 0.5).bfloat16()
    resid = torch.randn(M, H, device=dev, generator=g).bfloat16()

    def bench(fn, n=50, reps=10):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn(0)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for i in range(n):
                    fn(i)
            for _ in range(reps):
                gr.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                gr.replay()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * n)

    for name, N, K in (("qkv", QKV, H), ("o", H, nq * d), ("gate_up", 2 * I, H), ("down", H, I)):
        ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g) for _ in range(copies)]
        wps = [ops.pack_weight_i8(w) for w in ws]
        w_s = torch.rand(N, device=dev, generator=g) * 0.0002 + 0.0001
        a8 = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        if name == "qkv":
            bias = torch.randn(N, device=dev, generator=g).bfloat16()
            fn = lambda i: ops.scaled_matmul_rope_cache(a8, wps[i % copies], a_scale, w_s, bias, pos, cos_sin, slots, kc, vc, nq, nkv, d)
            what = "scaled_matmul_rope_cache_packed: GEMM -> slabs -> dequant + RoPE + KV write (two launches)"
        elif name == "gate_up":
            fn = lambda i: ops.scaled_matmul_silu_mul_quant(a8, ws[i % copies], a_scale, w_s, torch.bfloat16, None, b_packed=wps[i % copies])
            what = "scaled_matmul_gate_up_act + quantize_with_row_amax (two launches)"
        else:
            fn = lambda i: ops.scaled_matmul_add_rms_norm(a8, ws[i % copies], a_scale, w_s, resid, norm_w, 1e-6, None, quantize=True,
                                                          b_packed=wps[i % copies])
            what = "scaled_matmul_add_rms_norm_packed: GEMM -> slabs -> add + RMSNorm + int8 quant (two launches)"
        if fn(0) is None:
            out[name] = {"error": "the fused form declined this shape"}
            continue
        us = bench(fn)
        byts = N * K + M * K
        floor = max(byts / HBM_PEAK_GBS / 1e3, 2.0 * M * N * K / 5000.0 / 1e6)
        out[name] = {"us": round(us, 1), "what": what, "weight_gbs": round(N * K / us / 1e3, 1),
                     "frac_of_hbm": round(byts / us / 1e3 / HBM_PEAK_GBS, 4), "tops": round(2.0 * M * N * K / us / 1e6, 1),
                     "floor_us": round(floor, 1)}
        total += us
        del ws, wps
        torch.cuda.empty_cache()
    out["sum_us"] = round(total, 1)
    return out


def via_shim_leg(model, margs, md, kv_caches, tokens, positions, steps, warmup):
    """The decode step through the drop-in boundary itself (round-2 review, missing #6): every operator goes through
    shim/xllm_mi355_shim (xllm::kernel::mi355::* with the reference's torch::Tensor signatures + layer::AttentionImpl) in the
    reference's operator ORDER -- apply_norm, w8a8 linear = scaled_quantize + scaled_matmul, rotary_embedding,
    AttentionImpl::forward (KV write + paged attention), ... (qwen2_decoder_layer.cpp:87-110, qwen2_attention.cpp:132-193,
    dense_mlp.cpp:97-116, linear.cpp:481-507) -- WITHOUT the cross-operator N1 fusions, which have no reference operator to hide
    behind. Weights are packed at load time through the shim (pack_w8a8_weight); the step is replayed from one HIP graph."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import build_shim
    spec = importlib.util.spec_from_file_location("xllm_mi355_shim", build_shim.main())
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    eps = margs.rms_norm_eps
    for L in model.layers:
        for lin in (L.qkv_proj, L.o_proj, L.gate_up_proj, L.down_proj):
            m.pack_w8a8_weight(lin.weight)                      # the USE_MI355 branch of the loader
    slots, kv_lens, table, max_kv = md.slot_mapping, md.kv_seq_lens, md.block_table, md.max_seq_len

    def linear(lin, x):                                          # linear.cpp:481-507
        q8, sc = m.scaled_quantize(x)
        return m.scaled_matmul(q8, lin.weight, sc, lin.w_scale, lin.bias)

    def step():
        x = torch.nn.functional.embedding(tokens, model.embed)
        residual = None
        for L, kvc in zip(model.layers, kv_caches):
            if residual is None:                                 # apply_norm (qwen2_decoder_layer.cpp:66-85)
                residual = x
                h = torch.empty_like(x)
                m.rms_norm(h, x, L.input_norm_w, eps)
            else:
                m.fused_add_rms_norm(x, residual, L.input_norm_w, eps)
                h = x
            qkv = linear(L.qkv_proj, h)
            q, k, v = qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size], qkv[:, L.q_size + L.kv_size:]
            m.rotary_embedding(positions, q, k, model.cos_sin, True)
            attn = m.attention_forward(q, k, v, kvc.k_cache, kvc.v_cache, slots, kv_lens, table, L.nq, L.nkv, L.d, max_kv)
            x = linear(L.o_proj, attn.reshape(attn.size(0), -1))
            m.fused_add_rms_norm(x, residual, L.post_norm_w, eps)
            gate_up = linear(L.gate_up_proj, x)
            act = torch.empty(gate_up.size(0), gate_up.size(1) // 2, dtype=gate_up.dtype, device=gate_up.device)
            m.act_and_mul(act, gate_up, "silu")
            x = linear(L.down_proj, act)
        m.fused_add_rms_norm(x, residual, model.norm_w, eps)
        return torch.argmax(m.matmul(x, model.lm_head.weight, None), dim=-1)

    for _ in range(max(warmup, 2)):
        out_eager = step()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        step()                                                   # this stream's scratch buffers are created outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            out_graph = step()
    torch.cuda.current_stream().wait_stream(cap)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": round(ms, 4), "tokens_per_s": round(tokens.numel() / ms * 1e3, 2),
            "path": "shim/xllm_mi355_shim: reference operator order, no N1 fusions, weights packed at load, one HIP graph",
            "tokens_equal_eager_replay": bool(torch.equal(out_graph, out_eager)),
            "packed_weights": int(m.packed_weight_cache_size())}

def engine_leg(model, kv_caches, B, ctx, block_size, n_blocks, dev, steps, margs):
    """the same decode workload driven through the step-level harness (SURVEY 8f N2 + N3): per step the HOST builds the
    batch's indexing data, one H2D copy, device-side metadata refresh, ONE graph replay (model + lm_head + argmax) and
    the sampled tokens come back to the host -- the sequences really advance by one token per step. Reported beside the
    headline number (which times the graph alone): the difference is the per-step host work + the D2H sync."""
    from xllm_amd import engine
    warm = 2
    steps = max(steps, 20)     # (the greedy / random difference is ~0.2 ms: ten steps do not resolve it)
    pages = (ctx + warm + steps + block_size - 1) // block_size
    if B * pages > n_blocks:
        return None
    g = torch.Generator().manual_seed(4321)
    blocks = torch.randperm(n_blocks, generator=g)[: B * pages].view(B, pages).tolist()
    first = torch.randint(0, margs.vocab_size, (B,), generator=g).to(torch.int32).to(dev)
    def run(temperature, **kw):
        eng = engine.DecodeEngine(model, kv_caches, block_size, [ctx - 1] * B, blocks, first, ctx + warm + steps, temperature, **kw)
        for _ in range(warm):
            eng.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    el = run(0.0)
    out = {"ms_per_step": round(el / steps * 1e3, 4), "tokens_per_s": round(B * steps / el, 2), "steps": steps,
           "what": "xllm_amd.engine.DecodeEngine: host batch builder + H2D + device metadata refresh + graph replay + "
                   "argmax + D2H per step, sequences grow from ctx-1"}
    try:   # the random path of Sampler::forward (sampler.cpp:100-137): temperature 0.8, top-k 50, top-p 0.9 on every row
        el_r = run(0.8, top_k=50, top_p=0.9, seed=1)
        out["random_sampling"] = {"ms_per_step": round(el_r / steps * 1e3, 4), "over_greedy": round(el_r / el - 1.0, 4),
                                  "what": "the same steps with temperature 0.8 + top-k 50 + top-p 0.9 sampling: lm_head -> "
                                          "apply_top_k_top_p in place on the 16-bit logits -> softmax + random_sample in one launch "
                                          "(no fp32 copy of the logits, no [B, V] probabilities)"}
    except Exception as e:  # noqa: BLE001
        out["random_sampling"] = {"error": repr(e)}
    return out


def prefill_leg(model, margs, kv_caches, block_size, ctx, dev, world, tp_size, dp_size, sync_all):
    """prefill TFLOPS (second half of the BASELINE metric): one chunk of 2 x ctx tokens through the
    full model -- causal varlen flash attention, W8A8 GEMMs at M = 8192, KV written to fresh pages; logits only
    for the last token of each sequence (llm_model_base.h:193-204). flops = 2*T*sum(N*K) + 2*nq*d*S^2*L per seq."""
    from xllm_amd import attention, ops
    nseq = 2
    T = nseq * ctx
    pages = ctx // block_size
    table = torch.arange(nseq * pages, dtype=torch.int32).view(nseq, pages)
    bi = attention.build_batch_input([0] * nseq, [ctx] * nseq, table.tolist(), block_size)   # nothing cached: pure prefill
    md = attention.build_attention_metadata(bi, is_prefill=True, is_chunked_prefill=False, device=dev)
    tokens = torch.randint(0, margs.vocab_size, (T,), device=dev)
    positions = bi.positions.to(dev).long()
    last = torch.arange(ctx - 1, T, ctx, device=dev)

    def chunk():
        hidden = model.forward(tokens, positions, md, kv_caches)
        return model.greedy_tokens(hidden.index_select(0, last))

    chunk()
    sync_all()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        chunk()
    sync_all()
    dt = (time.perf_counter() - t0) / n
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    H, I, d = margs.hidden_size, margs.intermediate_size, margs.head_dim
    w_macs = (margs.n_heads + 2 * margs.n_kv_heads) * d * H + margs.n_heads * d * H + 2 * I * H + I * H
    flops = margs.n_layers * (2 * T * w_macs + nseq * 2 * margs.n_heads * d * ctx * ctx) + 2 * nseq * H * margs.vocab_size
    flops *= dp_size  # every DP replica prefills its own chunk
    peak = 5000.0 * world  # int8 dense MFMA, TOP/s per GPU
    return {"tflops": round(flops / dt / 1e12, 1), "ms_per_chunk": round(dt * 1e3, 3), "tokens": T * dp_size,
            "tokens_per_s": round(T * dp_size / dt, 1), "flops_per_chunk": flops, "mfma_peak_tflops": peak,
            "frac": round(flops / dt / 1e12 / peak, 4), "parallelism": f"tp{tp_size}" + (f"xdp{dp_size}" if dp_size > 1 else "")}


def micro(model, md, kv_caches, tokens, positions, B, fh):
    """per-operator timings of one layer (cuda events, 20 iterations each) -- tuning aid, not the metric"""
    from xllm_amd import ops
    L = model.layers[0]
    dev = tokens.device
    H, I = model.args.hidden_size, L.I
    x = torch.randn(B, H, device=dev).bfloat16()
    res = torch.randn(B, H, device=dev).bfloat16()
    qkv = torch.randn(B, L.q_size + 2 * L.kv_size, device=dev).bfloat16()
    gu = torch.randn(B, 2 * I, device=dev).bfloat16()
    kvc = kv_caches[0]

    def t(name, fn, bytes_=None, flops=None, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        extra = ""
        if bytes_:
            extra += f"  {bytes_ / us / 1e3:8.1f} GB/s"
        if flops:
            extra += f"  {flops / us / 1e6:8.1f} TOP/s"
        print(f"[micro] {name:34s} {us:9.1f} us{extra}", file=fh)

    q8, s8 = ops.scaled_quantize(x)
    t("rms_norm+int8 quant (fused add)", lambda: ops.rms_norm_dynamic_int8_quant(x, L.input_norm_w, 1e-6, residual=res),
      bytes_=B * H * (2 + 2 + 2 + 1))
    at = torch.randn(B, L.q_size, device=dev).bfloat16()
    act = torch.randn(B, I, device=dev).bfloat16()
    if L.qkv_proj.mode == "int8":
        qa, sa = ops.scaled_quantize(at)
        qd, sd = ops.act_and_mul_dynamic_int8_quant(gu)
        for nm, lin, (qi, si) in (("qkv", L.qkv_proj, (q8, s8)), ("gate_up", L.gate_up_proj, (q8, s8)), ("o", L.o_proj, (qa, sa)),
                                  ("down", L.down_proj, (qd, sd))):
            N, K = lin.weight.shape
            t(f"scaled_matmul {nm} [{B}x{N}x{K}]",
              lambda lin=lin, qi=qi, si=si: ops.scaled_matmul(qi, lin.weight, si, lin.w_scale, torch.bfloat16, lin.bias),
              bytes_=N * K + B * K + B * N * 2, flops=2 * B * N * K)
    else:  # 16-bit linears (cfg2): dcu::matmul
        for nm, lin, inp in (("qkv", L.qkv_proj, x), ("gate_up", L.gate_up_proj, x), ("o", L.o_proj, at), ("down", L.down_proj, act)):
            N, K = lin.weight.shape
            t(f"matmul {nm} [{B}x{N}x{K}]", lambda lin=lin, inp=inp: ops.matmul(inp, lin.weight, lin.bias),
              bytes_=(N * K + B * K + B * N) * 2, flops=2 * B * N * K)
    t("silu_mul+int8 quant", lambda: ops.act_and_mul_dynamic_int8_quant(gu), bytes_=B * I * 5)
    t("rope", lambda: ops.rotary_embedding(positions, qkv[:, :L.q_size], qkv[:, L.q_size:L.q_size + L.kv_size], model.cos_sin, True, head_size=L.d))
    q3 = qkv[:, :L.q_size].unflatten(-1, (L.nq, L.d))
    ctx = md.max_seq_len
    t("paged decode attention", lambda: ops.paged_attention(q3, kvc.k_cache, kvc.v_cache, None, md.kv_seq_lens, md.block_table, 1, ctx, L.attn.scale),
      bytes_=B * (ctx * L.nkv * L.d * 4 + 4 * L.nq * L.d))
    h = torch.randn(B, H, device=dev).bfloat16()
    V = model.lm_head.weight.shape[0]
    t(f"lm_head matmul [{B}x{V}x{H}]", lambda: ops.matmul(h, model.lm_head.weight), bytes_=V * H * 2, flops=2 * B * V * H, n=5)


if __name__ == "__main__":
    main()
