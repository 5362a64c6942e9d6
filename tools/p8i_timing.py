"""round 6: phases of ONE workgroup of the 8-phase int8 kernel (timing build -DP8I_TIMING, tools/build_ablations.sh p8itime): prologue
(entry -> first K tile readable), K loop, epilogue (last MFMA -> stores accepted), wall clock (100 MHz) and shader cycles"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops, _lib
q = lambda t: torch.round(t / (t.abs().amax(-1, keepdim=True) / 127.0)).to(torch.int8)
for name, M, N, K in (("gate_up", 8192, 37888, 3584), ("qkv", 8192, 4608, 3584), ("o", 8192, 3584, 3584), ("down", 8192, 3584, 18944),
                      ("moe_w13", 512, 1536, 2048), ("moe_w2", 512, 2048, 768)):
    w, a = q(torch.randn(N, K, device="cuda")), q(torch.randn(M, K, device="cuda"))
    a_s = torch.rand(M, device="cuda"); w_s = torch.rand(N, device="cuda")
    for _ in range(3):
        ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.scaled_matmul(a, w, a_s, w_s, torch.bfloat16)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    out = (ctypes.c_longlong * 16)()
    rc = _lib.lib().xllm_mi355_debug_p8i(out)
    pw, kw, ew, pc, kc, ec, nk = [out[i] for i in range(7)]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"[p8i timing] {name:8s} M={M} N={N} K={K}: launch {us:7.1f} us = {2.0 * M * N * K / us / 1e6:5.0f} TOP/s, {tiles} tiles = {tiles / 256:.2f} rounds | one workgroup: "
          f"prologue {pw / 100:5.2f} us ({pc} cyc) | K loop {kw / 100:6.2f} us ({kc / max(nk, 1):.0f} cyc / tile x {nk}, {kc / max(kw, 1) / 10:.2f} GHz) | "
          f"epilogue {ew / 100:5.2f} us ({ec} cyc) | sum {(pw + kw + ew) / 100:.2f} us")
