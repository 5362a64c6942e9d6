"""reads the in-kernel tile-loop timing of an ablation build (-DXM_ABL_PF_TIMING) of the LDS-DMA prefill kernel: shader
cycles and wall time per 64-key tile of one mid-grid workgroup -> the shader clock the kernel is granted"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops, _lib
NQ, NKV, D, S, nseq = 28, 4, 128, 4096, 2
T = S * nseq
qkv = torch.randn(T, (NQ + 2 * NKV) * D, device="cuda").bfloat16()
if os.environ.get("PF_ZERO"):
    qkv.zero_()
q = qkv[:, :NQ * D].unflatten(-1, (NQ, D)); k = qkv[:, NQ * D:(NQ + NKV) * D].unflatten(-1, (NKV, D)); v = qkv[:, (NQ + NKV) * D:].unflatten(-1, (NKV, D))
cu = torch.tensor([0, S, 2 * S], dtype=torch.int32, device="cuda")
for _ in range(8):
    ops.prefill_attention(q, k, v, cu, cu, S, 1.0 / math.sqrt(D), True)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 12)()
rc = _lib.lib().xllm_mi355_debug_pf(out)
cyc, wall, nt = out[0], out[1], out[2]
print(f"[pf timing] rc={rc}: {nt} tiles, {cyc / nt:.0f} shader cycles / tile, {wall * 10 / nt:.0f} ns / tile, clock {cyc / (wall * 10):.2f} GHz")
for grp in (0, 1):
    ph = out[4 + 4 * grp:8 + 4 * grp]
    if any(ph):
        print("[pf phases] group %d, cycles / tile: barrier wait + DMA issue %.0f | V reads + PV %.0f | K reads + QK^T %.0f | softmax %.0f" % ((grp,) + tuple(x / nt for x in ph)))
