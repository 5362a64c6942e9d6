"""gemm_one.py M N K -- a handful of launches of one W8A8 GEMM under rocprofv3 (kernel trace or --pmc passes).
GEMM_DIST=uniform (default) | gauss (what per-row int8 quantisation of Gaussian tensors yields) | zero (no operand toggling);
GEMM_PACKED=1: the weight-stream kernel on pre-packed weights; GEMM_GU=1: the gate_up form with SiLU * mul in the epilogue;
GEMM_LAUNCHES (default 6)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
dist = os.environ.get("GEMM_DIST", "uniform")


def make(r, c):
    if dist == "zero":
        return torch.zeros(r, c, dtype=torch.int8, device="cuda")
    if dist == "gauss":
        t = torch.randn(r, c, device="cuda")
        return torch.round(t / (t.abs().amax(-1, keepdim=True) / 127.0)).to(torch.int8)
    return torch.randint(-127, 128, (r, c), dtype=torch.int8, device="cuda")


ws = [make(N, K) for _ in range(2)]
a = make(M, K)
a_s = torch.rand(M, device="cuda") * 0.01; w_s = torch.rand(N, device="cuda") * 0.01
packed = os.environ.get("GEMM_PACKED") == "1"   # the weight-stream kernel on pre-packed weights (set XLLM_MI355_PACKED=1 too)
wps = [ops.pack_weight_i8(w) if packed else None for w in ws]
for i in range(int(os.environ.get("GEMM_LAUNCHES", "6"))):
    if os.environ.get("GEMM_GU") == "1":
        ops.scaled_matmul_silu_mul_quant(a, ws[i % 2], a_s, w_s, torch.bfloat16, None, b_packed=wps[i % 2])
    else:
        ops.scaled_matmul(a, ws[i % 2], a_s, w_s, torch.bfloat16, b_packed=wps[i % 2])
torch.cuda.synchronize()
