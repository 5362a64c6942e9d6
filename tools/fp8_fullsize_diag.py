#!/usr/bin/env python3
"""where does the fp8 GEMM differ from the fp64 product of the dequantised operands? (diagnostic for test_fp8_gemm_full_size)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops  # noqa: E402

DEV = "cuda"
for M, N, K, per in [(128, 3072, 1536, True), (128, 7168, 2048, False), (256, 3584, 18944, True), (256, 37888, 3584, True)]:
    gd = torch.Generator(device=DEV).manual_seed(M * 7 + N)
    a = (torch.randn(M, K, device=DEV, generator=gd) * 2).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, device=DEV, generator=gd) * 0.5).to(torch.float8_e4m3fn)
    a_s = torch.rand(M if per else 1, device=DEV, generator=gd) * 0.05 + 0.01
    w_s = torch.rand(N if per else 1, device=DEV, generator=gd) * 0.02 + 0.01
    out = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.bfloat16)
    a64 = a.double() * (a_s.double()[:, None] if per else a_s.double())
    w64 = w.double() * (w_s.double()[:, None] if per else w_s.double())
    ref = a64 @ w64.T
    mag = a64.abs() @ w64.abs().T
    t32 = ((a.float() @ w.float().T) * (a_s[:, None] if per else a_s) * (w_s[None, :] if per else w_s))
    for name, o in (("hip bf16", out), ("torch fp32 product -> bf16", t32.bfloat16()), ("fp64 -> bf16", ref.bfloat16())):
        err = (o.double() - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 2.0 ** -20 * mag
        bad = err > tol
        i = torch.argmax((err - tol).flatten())
        print(f"[{M}x{N}x{K} {'tc' if per else 'tensor'}] {name:28s} violations {int(bad.sum()):8d} / {bad.numel()}  "
              f"worst: err {err.flatten()[i]:.3e} ref {ref.flatten()[i]:.3e} out {float(o.flatten()[i]):.3e} mag {mag.flatten()[i]:.3e} "
              f"rel_l2 {((o.double() - ref).norm() / ref.norm()).item():.2e}")
    try:
        o32 = ops.fp8_scaled_matmul(a, w, a_s, w_s, torch.float32)
        err = (o32.double() - ref).abs()
        print(f"    fp32 output: max err/mag {(err / mag).max().item():.3e}  rel_l2 {((o32.double() - ref).norm() / ref.norm()).item():.2e}")
    except Exception as e:
        print("    fp32 output not supported:", str(e)[:80])
