#!/usr/bin/env python3
"""Free-running W8A8 parity, row by row (round-4 review, weak #1): two Qwen2-7B-geometry layers, one ragged decode step of 32
sequences. Prints, for the HIP path and for every control of tests/_model_parity.py (the oracle with its decode attention split
over the keys in 1 / 2 / 4 / 8 parts, softmax in base e or base 2 -- equally valid fp32 evaluation orders), the per-row logits
error against the oracle and WHERE each row first parts from the oracle: the first operator whose output differs and the first
per-token quantiser whose int8 codes differ (how many codes, by how much, whether the row's scale moved).
Test infrastructure (the oracle is the checker)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _model_parity as mp  # noqa: E402
from oracle import model as omodel  # noqa: E402

if __name__ == "__main__":
    seeds = [int(a) for a in sys.argv[1:] if a != '-v'] or [23]
    for seed in seeds:
        r = mp.int8_jump_experiment(seed=seed)
        B = len(r["lens"])
        print(f"==== weights seed {seed}, {B} rows; jump = logits rel err > {mp.JUMP:.0e}")
        for name, e in r["errors"].items():
            jumps = [i for i in range(B) if e[i] > mp.JUMP]
            print(f"{name:18s} jumps {len(jumps):2d}/{B}  max {max(e):.2e}  layer-0 attention elements differing {r['attn0_diff_elements'][name]:4d}  rows {jumps}")
            for i in (jumps if "-v" in sys.argv else []):
                print(f"      row {i:2d} len {r['lens'][i]:4d} err {e[i]:.2e}  {r['attribution'][name][i]}")
