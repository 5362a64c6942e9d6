import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda") for _ in range(2)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
a_s = torch.rand(M, device="cuda"); w_s = torch.rand(N, device="cuda")
packed = os.environ.get("GEMM_PACKED") == "1"   # the weight-stream kernel on pre-packed weights (set XLLM_MI355_PACKED=1 too)
wps = [ops.pack_weight_i8(w) if packed else None for w in ws]
for i in range(6):
    ops.scaled_matmul(a, ws[i % 2], a_s, w_s, torch.bfloat16, b_packed=wps[i % 2])
torch.cuda.synchronize()
