import torch, sys
sys.path.insert(0, '.')
from xllm_amd import ops, _lib
DEV = 'cuda'
ops._PACKED_16_POLICY = "1"
for dtype in (torch.bfloat16, torch.float16):
    for (M, I, K) in [(1, 128, 512), (17, 384, 1024), (128, 1280, 512)]:
        g = torch.Generator().manual_seed(3 * M + I + K)
        N = 2 * I
        a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(DEV)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
        wp = ops.pack_weight_16(w)
        _lib.lib().xllm_mi355_debug_ws_plan(0, 1)
        gate_up = ops.matmul(a, w, None, b_packed=wp)
        _lib.lib().xllm_mi355_debug_ws_plan(0, 0)
        ref = torch.empty(M, I, dtype=dtype, device=DEV)
        ops.act_and_mul(ref, gate_up, "silu")
        out = ops.matmul_silu_mul(a, w, None, b_packed=wp)
        d = (out.float() - ref.float()).abs()
        bad = (out != ref).nonzero()
        print(dtype, M, I, K, "mismatches", bad.size(0), "max", float(d.max()))
        for i in bad[:6].tolist():
            r, c = i
            print("   at", r, c, "fused", float(out[r, c]), "ref", float(ref[r, c]), "gate", float(gate_up[r, c]), "up", float(gate_up[r, I + c]))
