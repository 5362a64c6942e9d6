#!/usr/bin/env python3
"""round 6: the two grouped W8A8 GEMMs of the cfg5 slice (Qwen3-MoE, one EP-8 rank: 16 local experts, ~8.1 k rows, H 2048, moe_I 768)
by themselves: N launches captured in ONE HIP graph (no host gaps), weights rotated over copies. Prints us per launch and TOP/s.
GG_ROWS_PER_EXPERT: rows per expert (default: the routing of a random gate, ~506)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
H, I, E_local, topk, T = 2048, 768, 16, 8, 8192
g = torch.Generator(device=dev).manual_seed(0)
rpe = os.environ.get("GG_ROWS_PER_EXPERT")
if rpe:
    sizes = torch.full((E_local,), int(rpe), dtype=torch.int32, device=dev)
else:
    ids = torch.randint(0, 128, (T * topk,), device=dev, generator=g)
    sizes = torch.bincount(ids, minlength=128)[:E_local].to(torch.int32)
rows = int(sizes.sum())
x = torch.randint(-127, 128, (T, H), dtype=torch.int8, device=dev, generator=g)
xs = torch.rand(T, device=dev, generator=g) * 0.01 + 0.001
dst_src = torch.randint(0, T * topk, (rows,), dtype=torch.int32, device=dev, generator=g)
NC = 4
w13 = [torch.randint(-127, 128, (E_local, 2 * I, H), dtype=torch.int8, device=dev, generator=g) for _ in range(NC)]
w13s = torch.rand(E_local, 2 * I, device=dev, generator=g) * 0.01 + 0.001
w2 = [torch.randint(-127, 128, (E_local, H, I), dtype=torch.int8, device=dev, generator=g) for _ in range(NC)]
w2s = torch.rand(E_local, H, device=dev, generator=g) * 0.01 + 0.001
a2 = torch.randint(-127, 128, (rows, I), dtype=torch.int8, device=dev, generator=g)
a2s = torch.rand(rows, device=dev, generator=g) * 0.01 + 0.001


def graph_time(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(n):
                fn(i)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gr.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


us13 = graph_time(lambda i: ops.group_gemm_w8a8(x, xs, w13[i % NC], w13s, sizes, torch.bfloat16, row_index=dst_src, index_div=topk))
us2 = graph_time(lambda i: ops.group_gemm_w8a8(a2, a2s, w2[i % NC], w2s, sizes, torch.bfloat16))
f13, f2 = 2.0 * rows * 2 * I * H, 2.0 * rows * H * I
print(f"[ggemm] rows {rows} (per expert {int(sizes.min())}..{int(sizes.max())}): w13 (gather) {us13:7.1f} us = {f13 / us13 / 1e6:6.0f} TOP/s "
      f"({f13 / us13 / 1e6 / 50:4.1f} % of 5 POP/s) | w2 {us2:7.1f} us = {f2 / us2 / 1e6:6.0f} TOP/s ({f2 / us2 / 1e6 / 50:4.1f} %)   "
      f"[plan kernel included in each launch]")
