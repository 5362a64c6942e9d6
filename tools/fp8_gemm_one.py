import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
ws = []
for _ in range(2):
    w = torch.randn(N, K, device="cuda")
    ws.append((w / (w.abs().max() / 448.0)).to(torch.float8_e4m3fn))
a, a_s = ops.fp8_scaled_quantize(torch.randn(M, K, device="cuda").bfloat16())
w_s = torch.tensor([0.01], device="cuda")
for i in range(int(__import__("os").environ.get("GEMM_LAUNCHES", "4"))):
    ops.fp8_scaled_matmul(a, ws[i % 2], a_s, w_s, torch.bfloat16)
torch.cuda.synchronize()
