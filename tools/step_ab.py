"""step_ab.py -- in-process A/B of the decode step: ONE model, one HIP graph per variant, timed interleaved.

Two `bench.py` processes on the same box differ by +-0.3 ms per step (clock / HBM state), which is more than most of the
changes of round 3 are worth; here every variant is captured from the same weights and KV caches and the replays alternate
(rounds x (variants x replays)), so that box and clock drift hit all variants alike.

    python tools/step_ab.py [B] [ctx] [variants...]      variants: name=setting;setting  with settings
        fuse_gu=0|1        ops._GATE_UP_FUSION
        greedy=0|1         ops._GREEDY_FUSION (lm_head + argmax in one pass, round 4)
        packed=auto|0|1    ops._PACKED_POLICY
        ws_ng=N  ws_sl=N  ws_rows=128|256   xllm_mi355_gemm_plan_hint (product API, thread-local)
        ws_waves=N         xllm_mi355_debug_ws_waves (80, 81, 140..142, 150 | 151 = slab stores plain | non-temporal, 0)   [tuning flavour]
        shape=N,K,ng,sl    xllm_mi355_debug_ws_plan_shape: tile width / K slices of ONE GEMM of the step
        attn=s,h,d,e       xllm_mi355_debug_decode_plan: split-KV count, kv heads per workgroup, deep prefetch, exclusive CU
        idle=before|after,US   an idle gap of US microseconds in front of / behind every decode-attention launch (analysis)
        env:NAME=VALUE     os.environ (only for switches that are read at call time)
    default: python tools/step_ab.py 256 4096 fused=fuse_gu=1 unfused=fuse_gu=0

The xllm_mi355_debug_* entry points exist only in the tuning flavour of the library (round 4):
    make -C xllm_amd/csrc tuning && XLLM_MI355_LIB=$PWD/xllm_amd/lib/libxllm_mi355_tuning.so python tools/step_ab.py ...
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xllm_amd import _lib, layers, ops  # noqa: E402
from xllm_amd.attention import KVCache  # noqa: E402
import bench  # noqa: E402


def apply(settings):
    ng = sl = rows = 0
    for s in settings:
        if not s:
            continue
        k, v = s.split("=", 1)
        if k == "fuse_gu":
            ops._GATE_UP_FUSION = v == "1"
        elif k == "greedy":
            ops._GREEDY_FUSION = v
        elif k == "ws_ng":
            ng = int(v)
        elif k == "ws_sl":
            sl = int(v)
        elif k == "ws_rows":
            rows = int(v)
        elif k == "ws_waves":
            _lib.lib().xllm_mi355_debug_ws_waves(int(v))
        elif k == "shape":                       # shape=N,K,ng,slices : plan override for that GEMM only
            n_, k_, g_, s_ = (int(x) for x in v.split(","))
            import ctypes
            _lib.lib().xllm_mi355_debug_ws_plan_shape(ctypes.c_longlong(n_), ctypes.c_longlong(k_), g_, s_)
        elif k == "attn":                        # attn=splits,hpw,deep,excl (0 = planner / default)
            _lib.lib().xllm_mi355_debug_decode_plan(*(int(x) for x in v.split(",")))
        elif k == "packed":                      # packed=auto|r2|0|1 : ops._PACKED_POLICY (which decode GEMMs run on packed weights)
            ops._PACKED_POLICY = v
        elif k == "idle":                        # idle=before|after,US : an idle gap of US microseconds next to every decode-attention launch
            where, us = v.split(",")
            IDLE["where"], IDLE["us"] = where, float(us)
        elif k.startswith("env:"):
            os.environ[k[4:]] = v
        else:
            raise SystemExit(f"unknown setting {s}")
    _lib.lib().xllm_mi355_gemm_plan_hint(ng, sl, rows)


IDLE = {"where": None, "us": 0.0}
_orig_attn = ops.paged_decode_attention_int8


def _attn_with_idle(*a, **kw):
    import ctypes
    st = torch.cuda.current_stream().cuda_stream
    if IDLE["where"] == "before":
        _lib.lib().xllm_mi355_debug_idle(ctypes.c_double(IDLE["us"]), ctypes.c_void_p(st))
    out = _orig_attn(*a, **kw)
    if IDLE["where"] == "after":
        _lib.lib().xllm_mi355_debug_idle(ctypes.c_double(IDLE["us"]), ctypes.c_void_p(st))
    return out


ops.paged_decode_attention_int8 = _attn_with_idle


def reset():
    IDLE["where"], IDLE["us"] = None, 0.0
    ops._PACKED_POLICY = "auto"
    ops._GATE_UP_FUSION = True
    ops._GREEDY_FUSION = "auto"
    import ctypes
    _lib.lib().xllm_mi355_gemm_plan_hint(0, 0, 0)
    if hasattr(_lib.lib(), "xllm_mi355_debug_ws_waves"):      # tuning flavour only
        _lib.lib().xllm_mi355_debug_ws_plan_shape(ctypes.c_longlong(0), ctypes.c_longlong(0), 0, 0)
        _lib.lib().xllm_mi355_debug_ws_waves(0)
        _lib.lib().xllm_mi355_debug_decode_plan(0, 0, 0, 0)


def main():
    argv = sys.argv[1:]
    B = int(argv[0]) if argv else 256
    ctx = int(argv[1]) if len(argv) > 1 else 4096
    specs = argv[2:] or ["fused=fuse_gu=1", "unfused=fuse_gu=0"]
    mode = os.environ.get("STEP_MODE", "int8")
    rounds, replays = int(os.environ.get("STEP_ROUNDS", "6")), int(os.environ.get("STEP_REPLAYS", "20"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    margs = layers.ModelArgs.qwen2_7b()
    dtype = torch.bfloat16
    model = layers.Qwen2Model(margs, mode, dtype, dev, seed=1234)
    md, n_blocks = bench.build_metadata(B, ctx, 128, dev, seed=0)
    gen = torch.Generator(device=dev).manual_seed(99)
    kv = []
    for _ in model.layers:
        kc = torch.empty(n_blocks, 128, margs.n_kv_heads, margs.head_dim, dtype=dtype, device=dev).normal_(generator=gen)
        vc = torch.empty(n_blocks, 128, margs.n_kv_heads, margs.head_dim, dtype=dtype, device=dev).normal_(generator=gen)
        kv.append(KVCache(kc, vc))
    tokens = torch.randint(0, margs.vocab_size, (B,), device=dev, generator=gen)
    positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)

    def step():
        return model.greedy_tokens(model.forward(tokens, positions, md, kv))

    graphs, outs = [], []
    for spec in specs:
        name, _, rest = spec.partition("=")
        reset()
        apply(rest.split(";"))
        for _ in range(2):
            out = step()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            step()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                out = step()
        torch.cuda.current_stream().wait_stream(st)
        g.replay()
        torch.cuda.synchronize()
        graphs.append((name, g))
        outs.append(out.clone())
    reset()
    for (n, _), o in zip(graphs, outs):
        same = bool(torch.equal(o, outs[0]))
        print(f"[step_ab] variant {n}: tokens {'==' if same else '!='} variant {graphs[0][0]}")
    ms = {n: [] for n, _ in graphs}
    for r in range(rounds + 1):
        for n, g in graphs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(replays):
                g.replay()
            torch.cuda.synchronize()
            if r:                                   # round 0 warms the clocks
                ms[n].append((time.perf_counter() - t0) / replays * 1e3)
    base = sum(ms[graphs[0][0]]) / rounds
    for n, _ in graphs:
        v = ms[n]
        mean = sum(v) / len(v)
        print(f"[step_ab] B={B} ctx={ctx} {mode} {n:>14s}: {mean:7.3f} ms/step  (min {min(v):.3f} max {max(v):.3f})  {mean - base:+.3f} vs {graphs[0][0]}")


if __name__ == "__main__":
    main()
