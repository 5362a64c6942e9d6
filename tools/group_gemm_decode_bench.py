"""round 6: the grouped 16-bit GEMMs of a MoE DECODE step (a handful of rows per expert: gemm_wsb.hip grouped form) -- cfg4 shapes:
256 experts, 4 rows each, w13 [E, 512, 7168] (1.88 GB) and w2 [E, 7168, 256] (0.94 GB); bytes / time = the weight stream's rate"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
E, rows_per = 256, int(os.environ.get("GG_ROWS", "4"))
dev = "cuda"

def graph_time(fn, n=10, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

for N, K in ((512, 7168), (7168, 256), (512, 2048), (2048, 512)):
    w = (torch.randn(E, N, K, device=dev) * 0.02).bfloat16()
    x = torch.randn(E * rows_per, K, device=dev).bfloat16()
    if os.environ.get("GG_SKEW", "0") == "1":   # a decode step's routing under the group-limited gate: ~150 of 256 experts hold rows, up to ~40
        g = torch.Generator().manual_seed(3)
        probs = torch.rand(E, generator=g) ** 3
        pick = torch.multinomial(probs, E * rows_per, replacement=True, generator=g)
        cnt = torch.bincount(pick, minlength=E).to(torch.int32).to(dev)
    else:
        cnt = torch.full((E,), rows_per, dtype=torch.int32, device=dev)
    us = graph_time(lambda: ops.group_gemm(x, w, cnt))
    gb = E * N * K * 2 / 1e9
    tag = f"skewed (experts with rows {int((cnt > 0).sum())}, max {int(cnt.max())})" if os.environ.get("GG_SKEW", "0") == "1" else f"rows/expert={rows_per}"
    print(f"[group gemm decode] E={E} {tag} N={N} K={K}: {us:.1f} us  weights {gb:.2f} GB -> {gb / us * 1e3:.2f} TB/s ({gb / us * 1e3 / 8:.3f} of 8 TB/s)")
