"""round 6: per-phase shader cycles of one wave of flash_prefill_m32_kernel (timing build -DPF32_TIMING -DXM_TUNING), 2 x 4096 causal"""
import ctypes, math, os, sys, torch
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
out = (ctypes.c_longlong * 16)()
rc = _lib.lib().xllm_mi355_debug_pf32(out)
ph = [out[i] for i in range(4)]
tot, wall, nt = out[4], out[5], out[6]
print(f"[pf32 timing] of the first phase: own DMA slices landing (vmcnt) {out[7]/max(nt,1):.0f} | workgroup barrier {out[8]/max(nt,1):.0f} | K DMA issue {out[9]/max(nt,1):.0f} | V DMA issue {out[10]/max(nt,1):.0f} | up to the first fragment read {ph[0]/max(nt,1):.0f} cycles per tile")
print(f"[pf32 timing] rc={rc} tiles {nt}: per tile  barrier+DMA {(ph[0]+out[7]+out[8]+out[9]+out[10])/max(nt,1):.0f} | QK^T {ph[1]/max(nt,1):.0f} | mask+max+P0 {ph[2]/max(nt,1):.0f} | PV+P1..3 {ph[3]/max(nt,1):.0f} "
      f"| sum {(sum(ph)+out[7]+out[8]+out[9]+out[10])/max(nt,1):.0f} cycles  (loop {tot/max(nt,1):.0f} cycles/tile, clock {tot/max(wall,1)/10:.2f} GHz; each mark drains the LDS queue)")
