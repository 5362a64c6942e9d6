#!/usr/bin/env python3
"""round 6: what the random-sampling tail of the decode step costs at the headline shape ([256, 152064] logits), stage by stage:
the engine's unfused sequence (float() -> apply_top_k_top_p -> torch.softmax -> random_sample) against the fused sampler."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

dev = "cuda"
B, V = int(os.environ.get("SB_B", "256")), int(os.environ.get("SB_V", "152064"))
torch.manual_seed(0)
STD = float(os.environ.get("SB_STD", "3.0"))   # 0.5: the narrow logits of a random-init model (almost every column in two exponent bins)
logits16 = (torch.randn(B, V, device=dev) * STD).bfloat16()
temps = torch.full((B,), 0.8, device=dev)
top_k = torch.full((B,), 50, dtype=torch.int64, device=dev)
top_p = torch.full((B,), 0.9, device=dev)
u = torch.rand(B, device=dev)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"[sampler] B={B} V={V} std={STD}")
print(f"[sampler] float():            {timeit(lambda: logits16.float()):8.1f} us")
l32 = logits16.float()
for name, k, p in (("top_k only", top_k, None), ("top_p only", None, top_p), ("both", top_k, top_p), ("temperature only", None, None)):
    print(f"[sampler] apply_top_k_top_p fp32 ({name}): {timeit(lambda: ops.apply_top_k_top_p(l32.clone(), temps, k, p)) - timeit(lambda: l32.clone()):8.1f} us")
    print(f"[sampler] apply_top_k_top_p bf16 ({name}): {timeit(lambda: ops.apply_top_k_top_p(logits16.clone(), temps, k, p)) - timeit(lambda: logits16.clone()):8.1f} us")
print(f"[sampler] torch.softmax fp32:  {timeit(lambda: torch.softmax(l32, -1)):8.1f} us")
probs = torch.softmax(l32, -1)
print(f"[sampler] random_sample:       {timeit(lambda: ops.random_sample(probs, uniform=u)):8.1f} us")
print(f"[sampler] greedy_argmax bf16:  {timeit(lambda: ops.greedy_argmax(logits16)):8.1f} us")


def unfused():
    l = logits16.float()
    ops.apply_top_k_top_p(l, temps, top_k, top_p)
    return ops.random_sample(torch.softmax(l, -1), uniform=u)


print(f"[sampler] unfused sequence (both): {timeit(unfused):8.1f} us")
if hasattr(ops, "sample_top_k_top_p"):
    for name, k, p in (("top_k only", top_k, None), ("top_p only", None, top_p), ("both", top_k, top_p), ("temperature only", None, None)):
        # (in place: every call gets a fresh copy of the logits, whose cost is taken off)
        c16, c32 = timeit(lambda: logits16.clone()), timeit(lambda: l32.clone())
        print(f"[sampler] fused sample_top_k_top_p bf16 ({name}): {timeit(lambda: ops.sample_top_k_top_p(logits16.clone(), temps, k, p, uniform=u)) - c16:8.1f} us")
        print(f"[sampler] fused sample_top_k_top_p fp32 ({name}): {timeit(lambda: ops.sample_top_k_top_p(l32.clone(), temps, k, p, uniform=u)) - c32:8.1f} us")
    print(f"[sampler] softmax_random_sample bf16: {timeit(lambda: ops.softmax_random_sample(logits16, uniform=u)):8.1f} us   fp32: {timeit(lambda: ops.softmax_random_sample(l32, uniform=u)):8.1f} us")
