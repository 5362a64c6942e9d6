#!/usr/bin/env python3
"""Assembles profiles/r02_* from the raw outputs of this round's GPU runs (gpurun_out/r02*/, scratch, not tracked): one file per
topic, a header saying which command produced each block, long environment tags shortened. Re-run after a GPU call."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def read(rel):
    p = os.path.join(G, rel)
    if not os.path.exists(p):
        return None
    txt = open(p, errors="replace").read()
    txt = "\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l and "[W9" not in l and "[Gloo]" not in l)
    txt = re.sub(r"XLLM_MI355_LIB=\S*libxllm_mi355_ws_(\w+)\.so", r"build=\1", txt)
    return txt.strip() + "\n"


def write(name, blocks):
    out = []
    for title, rel in blocks:
        body = read(rel) if rel else None
        if rel and body is None:
            continue
        out.append("# " + title)
        if body:
            out.append(body)
    if out:
        with open(os.path.join(P, name), "w") as f:
            f.write("\n".join(out))
        print("wrote", name)


def json_line(rel):
    t = read(rel)
    if not t:
        return None
    for l in reversed(t.splitlines()):
        if l.startswith("{"):
            return json.loads(l)
    return None


def copy_json(rel, name):
    d = json_line(rel)
    if d is not None:
        with open(os.path.join(P, name), "w") as f:
            json.dump(d, f, indent=1)
            f.write("\n")
        print("wrote", name)


def main():
    os.makedirs(P, exist_ok=True)
    final = "r02final"
    copy_json(f"{final}/bench.json", "r02_bench.json")
    copy_json(f"{final}/bench_cfg2.json", "r02_bench_cfg2.json")
    copy_json(f"{final}/bench_cfg4_slice.json", "r02_bench_cfg4_slice.json")
    copy_json("r02small/bench_cfg5_slice.json", "r02_bench_cfg5_slice.json")   # (the slice gained its attention block after the closing run)
    for src, dst in ((f"{final}/bench_kernel_stats.txt", "r02_bench_kernel_stats.txt"),
                     (f"{final}/cfg2_kernel_stats.txt", "r02_bench_cfg2_kernel_stats.txt"),
                     (f"{final}/cfg4-slice_kernel_stats.txt", "r02_bench_cfg4_slice_kernel_stats.txt"),
                     ("r02small/cfg5-slice_kernel_stats.txt", "r02_bench_cfg5_slice_kernel_stats.txt"),
                     (f"{final}/pmc_fetch_size.txt", "r02_bench_pmc_fetch_size.txt"),
                     (f"{final}/pmc_write_size.txt", "r02_bench_pmc_write_size.txt"),
                     ("r02last/pytest.txt", "r02_pytest_gpu.txt")):
        t = read(src)
        if t:
            open(os.path.join(P, dst), "w").write(t)
            print("wrote", dst)
    mp = os.path.join(G, "model_parity.jsonl")
    if os.path.exists(mp):
        shutil.copy(mp, os.path.join(P, "r02_model_parity.jsonl"))
    write("r02_attn_split_sweep.txt", [
        ("tools/attn_bench.py on 1 x MI355X, paged decode attention (bf16 KV, d = 128, block 128); 20 launches, 3 rotating KV copies.\n"
         "# cfg2 = configs[1] (B=64, S=2048); dpN = one data-parallel replica of N (B = 256 / N, S = 4096); tp1 = cfg3\n"
         "# (1) round-1 plan (4 kv heads per workgroup always) under forced grid-level split counts, last block = its own heuristic",
         "r02verify2/attn_splits.txt"),
        ("(2) round-2 plan (first block: heuristic = heads per workgroup 4 -> 2 -> 1 before grid splits), then forced HPW x SPLITS",
         "r02fail/attn_hpw.txt"),
        ("(3) A/B switches on the round-2 plan: XLLM_MI355_DECODE_EXCL (one workgroup per CU by LDS reservation), _DEEP (3 register stages)",
         "r02b3/attn_ab.txt"),
        ("(4) what a bare streaming loop reaches on launches of this size: tools/hbm_read_bench.hip small (256 MB per launch, 8 launches\n"
         "# over distinct regions of a 2.15 GB buffer)", "r02b3/hbm_short.txt"),
    ])
    write("r02_gemm_ws.txt", [
        ("tools/gemm_bench.py <M> int8 on 1 x MI355X (graph replay of 20 launches, rotating weight copies), Qwen2-7B layer shapes.\n"
         "# (1) row-major kernels (round-1 planner: skinny / 8-phase)", "r02ws/gemm_rowmajor.txt"),
        ("(2) packed weight-stream kernel gemm_ws.hip forced on every shape (XLLM_MI355_PACKED=1 GEMM_PACKED=1), planner's own (NG, slices)",
         "r02ws/gemm_packed.txt"),
        ("(3) M = 256, forced tile widths (XLLM_MI355_WS_NG)", "r02ws/gemm_sweep256.txt"),
        ("(4) forced widths at the small M", "r02ws/gemm_sweep_ng.txt"),
        ("(5) forced K-slice counts", "r02ws/gemm_sweep_slices.txt"),
        ("(6) ablation builds (tools/build_ws_ablations.sh): NOADMA = no activation DMA, NOWDMA = no weight DMA, NOCOMPUTE = no LDS reads /\n"
         "# MFMAs, NOMFMA = LDS reads but no MFMAs; M = 32 rows first, then M = 256", "r02ws/gemm_ablate.txt"),
    ])
    write("r02_gemm_wsb.txt", [
        ("tools/gemm_bench.py <M> bf16 on 1 x MI355X, Qwen2-7B layer shapes + lm_head: (1) weight-stream 16-bit kernel gemm_wsb.hip (final form: weights two steps ahead; forced up to M = 64 here, the default policy takes M <= 32)",
         "r02wsb2/gemm_bf16_wsb.txt"),
        ("(2) the tiled kernels of round 1 (XLLM_MI355_WSB=0)", "r02wsb/gemm_bf16_tiled.txt"),
        ("(3) M = 64, forced K-slice counts (XLLM_MI355_WSB_SLICES)", "r02wsb2/gemm_bf16_slices.txt"),
        ("(4) M = 32, forced K-slice counts", "r02wsb2/gemm_bf16_slices32.txt"),
        ("(5) first version of the kernel (plain double buffer, two workgroups per CU allowed by the slice rule): M = 16 / 32 / 64", "r02wsb/gemm_bf16_wsb.txt"),
    ])
    write("r02_prefill_p.txt", [
        ("tools/prefill_p_accuracy.py + tools/prefill_attn_one.py (2 x 4096 tokens, 28 / 4 heads, d = 128): P mode 1 = one RNE-rounded\n"
         "# 16-bit P per score (default), P mode 2 = hi + lo (XLLM_MI355_PREFILL_P=2)", "r02b1/prefill_p.txt"),
    ])
    write("r02_fp8_fullsize_diag.txt", [
        ("tools/fp8_fullsize_diag.py: fp8 e4m3 GEMM against the fp64 product of the dequantised operands; violations of\n"
         "# |err| <= 2^-8 |ref| + 2^-20 sum|a||w| (the bar was then set to 2^-17, tests/test_gpu_fullsize.py)", "r02fail2/fp8_diag.txt"),
    ])
    write("r02_oneshot_allreduce.txt", [
        ("tests/test_gpu_oneshot_allreduce.py (two processes on one GPU)", "r02b3/oneshot_pytest.txt"),
        ("bench.py --gpus 2 --backend gloo --layout tp --oneshot-allreduce (two ranks SHARING one GPU: functional run, the absolute\n"
         "# time is meaningless)", "r02b3/bench_tp2_oneshot.json"),
        ("the same over gloo all-reduce", "r02b3/bench_tp2_gloo.json"),
    ])
    # fusions across launch boundaries at small batch (one data-parallel replica of 8: 32 sequences)
    def dp_lines(rel):
        t = read(rel) or ""
        out = []
        for l in t.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                out.append(f"    per-replica batch {d['config'].get('per_gpu_batch', '?'):>4}  ms_per_step {d['ms_per_step']:8.4f}")
        return "\n".join(out)

    def grep(rel, pats):
        t = read(rel) or ""
        return "\n".join("    " + l[:150] for l in t.splitlines() if any(p in l for p in pats))

    pats = ("slab_rope", "rope_and_cache", "slab_epilogue", "merge_kernel", "finish_int8", "scaled_quantize_i8")
    if read("r02finish3/bench_dp.txt"):
        with open(os.path.join(P, "r02_fusions.txt"), "w") as f:
            f.write("# bench.py --emulate-dp {2,4,8} + rocprofv3 --kernel-trace of the dp8 replica (32 sequences), 1 x MI355X\n"
                    "# (1) first version: ONE workgroup per token for both fusions (GEMM slabs -> dequant + RoPE + KV write; split-KV\n"
                    "#     partials -> merge + int8 quantise): both launches are as slow as or slower than the pairs they replace\n"
                    + dp_lines("r02finish2/bench_dp.txt") + "\n  without the two fusions:\n" + dp_lines("r02finish2/bench_dp_nofusion.txt")
                    + "\n" + grep("r02finish2/dp8_kernel_stats.txt", pats) + "\n"
                    "# (2) qkv fusion re-gridded into independent work items (RoPE pairs / plain elements over blockIdx.y), the merge + quantise\n"
                    "#     fusion switched off (its row maximum is a whole-row dependency: it stays opt-in, XLLM_MI355_ATTN_FINISH=1)\n"
                    + dp_lines("r02finish3/bench_dp.txt") + "\n  without the qkv fusion (XLLM_MI355_QKV_ROPE=0):\n"
                    + dp_lines("r02finish3/bench_dp_nofusion.txt") + "\n" + grep("r02finish3/dp8_kernel_stats.txt", pats) + "\n"
                    "# before either fusion (closing run): slab epilogue 4.6 us + rope_and_cache 7.1 us, merge 4.9 us + scaled_quantize 4.75 us per layer\n")
        print("wrote r02_fusions.txt")
        t = read("r02finish3/dp8_kernel_stats.txt")
        if t:
            open(os.path.join(P, "r02_dp8_kernel_stats.txt"), "w").write(t)

    # layouts: per-replica steps measured on one GPU
    rows = []
    for rel, what in (("r02finish3/bench_dp.txt", "final (with the qkv -> RoPE -> KV-write fusion)"), (f"{final}/bench_dp.txt", "closing run"), ("r02step/bench_dp.txt", "with the packed weight-stream GEMM"),
                      ("r02b1/bench_dp.txt", "before the packed weight-stream GEMM")):
        t = read(rel)
        if not t:
            continue
        for l in t.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                rows.append(f"{what:42s} per-replica batch {d['config'].get('per_gpu_batch', '?'):>4}  ms_per_step {d['ms_per_step']:8.4f}  "
                            f"whole-job tokens/s {d['value']:10.1f}  attention launch {d['roofline']['avg_launch_ms'] * 1e3:6.1f} us "
                            f"({d['roofline']['frac']:.3f} of 8 TB/s)")
    if rows:
        with open(os.path.join(P, "r02_layouts.txt"), "w") as f:
            f.write("# bench.py --emulate-dp k (k = 2, 4, 8): ONE data-parallel replica of k on one GPU (batch 256 / k, no exchange -- the\n"
                    "# decode path shards by sequences); whole-job tokens/s = 256 / step. TP shard shapes: --emulate-tp (DESIGN.md section 5)\n")
            f.write("\n".join(rows) + "\n")
        print("wrote r02_layouts.txt")


if __name__ == "__main__":
    main()
