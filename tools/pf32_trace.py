"""round 6: per-workgroup timeline of flash_prefill_m32_kernel (trace build -DPF32_TRACE -DXM_TUNING: tools/build_pf32_timing.sh with
PF32_DEFS=-DPF32_TRACE PF32_TAG=trace), 2 x 4096 causal: where the launch's time goes OUTSIDE a wave's steady-state tile loop"""
import ctypes, math, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import ops, _lib
NQ, NKV, D, S, nseq = 28, 4, 128, 4096, 2
T = S * nseq
qkv = torch.randn(T, (NQ + 2 * NKV) * D, device="cuda").bfloat16()
q = qkv[:, :NQ * D].unflatten(-1, (NQ, D)); k = qkv[:, NQ * D:(NQ + NKV) * D].unflatten(-1, (NKV, D)); v = qkv[:, (NQ + NKV) * D:].unflatten(-1, (NKV, D))
cu = torch.tensor([0, S, 2 * S], dtype=torch.int32, device="cuda")
for _ in range(4):
    ops.prefill_attention(q, k, v, cu, cu, S, 1.0 / math.sqrt(D), True)
torch.cuda.synchronize()
QB = int(os.environ.get("PF_QB", "128"))
nb = NQ * nseq * (S // QB)
nb = (nb + 7) // 8 * 8
out = (ctypes.c_longlong * (nb * 8))()
rc = _lib.lib().xllm_mi355_debug_pf32_trace(out, nb)
a = np.frombuffer(out, dtype=np.int64).reshape(nb, 8).copy()
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
tick = 0.01  # us per 100-MHz tick
st, l0, l1, en, nt, hw, xcc, qb = [a[:, i] for i in range(8)]
dur = (en.max() - t0) * tick
print(f"[pf32 trace] rc={rc} {len(a)} workgroups, launch span {dur:.1f} us (first entry -> last exit)")
print(f"  entry of the LAST workgroup to start: {(st.max() - t0) * tick:.1f} us; first exit {(en.min() - t0) * tick:.1f} us")
busy = ((en - st) * tick).sum()
slots = 512 if QB == 128 else 256
print(f"  sum of workgroup lifetimes {busy:.0f} us = {busy / slots:.1f} us per slot ({slots} slots) -> occupancy of the span {busy / slots / dur:.3f}")
pro = ((l0 - st) * tick); epi = ((en - l1) * tick); loop = ((l1 - l0) * tick)
print(f"  per workgroup: prologue (entry -> first DMA issued) mean {pro.mean():.2f} us, loop mean {loop.mean():.2f} us ({(loop / np.maximum(nt, 1)).mean():.3f} us per tile; "
      f"by tile count: short (nt <= 8) {(loop[nt <= 8] / nt[nt <= 8]).mean():.3f}, long (nt >= 48) {(loop[nt >= 48] / nt[nt >= 48]).mean():.3f}), epilogue mean {epi.mean():.2f} us")
print(f"  tiles in total {nt.sum()}; loop time in total {loop.sum():.0f} us; prologue total {pro.sum():.0f} us; epilogue total {epi.sum():.0f} us")
# active workgroups over time
edges = np.linspace(0, dur, 27)
line = []
for i in range(26):
    lo, hi = edges[i] / tick + t0, edges[i + 1] / tick + t0
    ov = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    line.append(f"{ov:.0f}")
print("  resident workgroups over the span (26 bins):", " ".join(line))
# per XCD finish times
for x in sorted(set(xcc & 0xf)):
    m = (xcc & 0xf) == x
    print(f"  xcd {x}: {m.sum()} workgroups, tiles {nt[m].sum()}, last exit {(en[m].max() - t0) * tick:.1f} us")
# the tail: workgroups alive in the last 20 us of the span
late = np.argsort(en)[-24:]
print("  last 24 exits: (start us, exit us, tiles, qb, xcd, cu-ish hw id)")
for i in late:
    print(f"    {(st[i]-t0)*tick:7.1f} {(en[i]-t0)*tick:7.1f}  nt {nt[i]:3d}  qb {qb[i]:3d}  xcd {xcc[i] & 0xf}  hw {hw[i] & 0xfff:03x}")
# dispatch order: the first 10 and last 10 blocks by start time
o = np.argsort(st)
print("  qb of the first 24 workgroups to start:", [int(qb[i]) for i in o[:24]])
print("  qb of the last 24 workgroups to start:", [int(qb[i]) for i in o[-24:]])
x0 = (xcc & 0xf) == 0
o0 = np.argsort(st[x0])
print("  xcd 0 start order (qb):", [int(v) for v in qb[x0][o0]][:80], "...")
