"""round 6: gate_up at a TP-4 rank's shape (M = 128, N = 2 x 4736, K = 3584): the SiLU-mul-epilogue GEMM (no K slices possible: the activation needs
the full sums) + its quantising pass, against a K-sliced GEMM -> slabs -> ONE row-wise consumer of the same size (proxy: the add + RMSNorm consumer)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
torch.manual_seed(0)
M, K = int(os.environ.get("GU_M", "128")), 3584
dev = "cuda"

def graph_time(fn, n=20, reps=3):
    for _ in range(3):
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

for N in (9472, 37888):
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    a_s = torch.rand(M, device=dev) * 0.02 + 0.001
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev)
    w_s = torch.rand(N, device=dev) * 0.02 + 0.001
    wp = ops.pack_weight_i8(w)
    t_fused = graph_time(lambda: ops.scaled_matmul_silu_mul_quant(a, w, a_s, w_s, torch.bfloat16, None, b_packed=wp))
    line = f"[gate_up M={M} N={N}] SiLU-mul epilogue GEMM + quantise pass {t_fused:.1f} us"
    if N <= 16384:
        res = torch.randn(M, N, device=dev).bfloat16()
        nw = torch.ones(N, device=dev).bfloat16()
        t_slab = graph_time(lambda: ops.scaled_matmul_add_rms_norm(a, w, a_s, w_s, res, nw, 1e-6, None, True, b_packed=wp))
        line += f" | K-sliced GEMM -> slabs -> add + RMSNorm + quant consumer {t_slab:.1f} us"
    print(line)
