#!/usr/bin/env python3
"""Prints the teacher-forced per-operator errors of tests/_model_parity.py (two Qwen2-7B-geometry layers, one ragged decode
step) operator by operator, plus the free-running logits drift and its control. Test infrastructure (the oracle is the
checker). Usage: model_parity_diag.py [mode ...]   (16bit int8 fp8)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _model_parity as mp  # noqa: E402
from oracle import model as omodel  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from xllm_amd import attention, layers  # noqa: E402
from xllm_amd.attention import KVCache  # noqa: E402

DEV = "cuda"


def run(mode):
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=23, fuse=False)
    w = omodel.export_weights(model)
    trace = []
    om = omodel.OracleQwen2(args, w, torch.bfloat16, trace=trace)
    g = torch.Generator().manual_seed(9)
    bs, lens = 128, [1, 129, 700, 128, 333, 5, 1024, 257]
    B = len(lens)
    need = [(n + bs - 1) // bs for n in lens]
    perm = torch.randperm(sum(need) + 3, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    nb = sum(need) + 3
    init = [(torch.randn(nb, bs, 4, 128, generator=g).bfloat16(), torch.randn(nb, bs, 4, 128, generator=g).bfloat16())
            for _ in range(args.n_layers)]
    ids = torch.randint(0, args.vocab_size, (B,), generator=g)
    pos = torch.tensor([n - 1 for n in lens])
    md = orc.build_batch_metadata(lens, [1] * B, blocks, bs)
    kcs, vcs = [k.clone() for k, _ in init], [v.clone() for _, v in init]
    ref = om.logits(om.forward(ids, pos, md, kcs, vcs, "decode")).float()
    bi = attention.build_batch_input([n - 1 for n in lens], lens, blocks, bs)
    amd = attention.build_attention_metadata(bi, False, False, DEV)
    print(f"==== mode {mode}: teacher-forced per-operator error (max rel L2 over rows, fraction of differing elements)")
    mp.teacher_forced_errors(
        model, mp.split_trace(trace, args.n_layers), pos, "decode",
        lambda li: dict(md=amd, caches=KVCache(init[li][0].to(DEV), init[li][1].to(DEV)), k_after=kcs[li], v_after=vcs[li]),
        report=lambda name, e, neq: print(f"  {name:32s} {e:.3e}  differing {neq:.5f}"))
    for fuse in (False, True):
        m2 = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=23, fuse=fuse)
        caches = [KVCache(k.to(DEV), v.to(DEV)) for k, v in init]
        lg = m2.logits(m2.forward(ids.to(DEV), bi.positions.long().to(DEV), amd, caches)).float().cpu()
        print(f" free-running HIP (fuse={fuse}) logits vs oracle: {[f'{x:.1e}' for x in mp.rel_rows(lg, ref).tolist()]}")
    alt = mp.AltOrderOracle(args, w, torch.bfloat16)
    kc2, vc2 = [k.clone() for k, _ in init], [v.clone() for _, v in init]
    lg = alt.logits(alt.forward(ids, pos, md, kc2, vc2, "decode")).float()
    print(f" CONTROL oracle with another evaluation order vs oracle: {[f'{x:.1e}' for x in mp.rel_rows(lg, ref).tolist()]}")


if __name__ == "__main__":
    for m in (sys.argv[1:] or ["16bit", "int8", "fp8"]):
        run(m)
