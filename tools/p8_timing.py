"""reads the in-kernel K-loop timing of an ablation build (-DP8_ABL_TIMING): shader cycles and wall time per K tile"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops, _lib
M, N, K = [int(x) for x in sys.argv[1:4]]
if os.environ.get("GEMM_DIST", "uniform") == "gauss":
    q = lambda t: torch.round(t / (t.abs().amax(-1, keepdim=True) / 127.0)).to(torch.int8)
    w, a = q(torch.randn(N, K, device="cuda")), q(torch.randn(M, K, device="cuda"))
elif os.environ.get("GEMM_DIST") == "zero":
    w, a = torch.zeros(N, K, dtype=torch.int8, device="cuda"), torch.zeros(M, K, dtype=torch.int8, device="cuda")
else:
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
a_s = torch.rand(M, device="cuda"); w_s = torch.rand(N, device="cuda")
for _ in range(5):
    ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 4)()
rc = _lib.lib().xllm_mi355_debug_p8(out)
cyc, wall, nk = out[0], out[1], out[2]
print(f"[p8 timing] M={M} N={N} K={K} rc={rc}: {cyc / nk:.0f} shader cycles / K tile, {wall * 10 / nk:.0f} ns / K tile, clock {cyc / (wall * 10) :.2f} GHz")
