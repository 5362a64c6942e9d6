import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda") for _ in range(2)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
a_s = torch.rand(M, device="cuda"); w_s = torch.rand(N, device="cuda")
for i in range(4):
    ops.scaled_matmul(a, ws[i % 2], a_s, w_s, torch.bfloat16)
torch.cuda.synchronize()
