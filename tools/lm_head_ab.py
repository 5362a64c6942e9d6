"""lm_head tail A/B (round 4): [M, 3584] x [152064, 3584]^T on packed 16-bit weights, (a) logits + greedy_argmax, (b) fused argmax
epilogue. HIP events around each form, three weight copies rotated (3.3 GB: nothing survives in the Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, K = 152064, 3584
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
ws = [ops.pack_weight_16((torch.randn(N, K, device=dev, generator=g) / 60).bfloat16()) for _ in range(3)]
wrow = torch.empty(N, K, dtype=torch.bfloat16, device=dev)   # (shape carrier for ops.matmul)


def run(fn, n=30):
    for i in range(3):
        fn(ws[i % 3])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        ev[i][0].record()
        fn(ws[i % 3])
        ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return t[len(t) // 2], t[0]


for hint in ((0, 0, 0), (0, 1, 0)):
    _lib.lib().xllm_mi355_gemm_plan_hint(*hint)
    print("plan hint", hint)
    print("  logits only          %.1f us (min %.1f)" % run(lambda w: ops.matmul(a, wrow, None, b_packed=w)))
    print("  logits + argmax      %.1f us (min %.1f)" % run(lambda w: ops.greedy_argmax(ops.matmul(a, wrow, None, b_packed=w))))
    print("  fused argmax         %.1f us (min %.1f)" % run(lambda w: ops.matmul_argmax(a, w, N, None)))
_lib.lib().xllm_mi355_gemm_plan_hint(0, 0, 0)
