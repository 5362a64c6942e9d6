"""The end-to-end oracle model (oracle/model.py) pinned on CPU:
 * against the HuggingFace Qwen2 implementation (fixture tests/golden/qwen2_hf_tiny.npz made by make_hf_golden.py, and live
   when `transformers` is importable): composition, head mapping, RoPE, fused projections, residual stream;
 * self-consistency of the three attention phases through the whole model;
 * the quantised modes against the de-quantised 16-bit model (sanity of the scale plumbing)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import model as omodel  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from xllm_amd import layers  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "qwen2_hf_tiny.npz")


def _hf_fixture(dtype=torch.float32):
    z = np.load(GOLD)
    c = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg.")}
    args = layers.ModelArgs(c["hidden_size"], c["num_hidden_layers"], c["num_attention_heads"], c["num_key_value_heads"],
                            c["hidden_size"] // c["num_attention_heads"], c["intermediate_size"], c["vocab_size"],
                            c["rms_norm_eps"], c["rope_theta"], c["max_position_embeddings"])
    t = lambda k: torch.from_numpy(z[k]).to(dtype)
    lin = lambda w, b=None: dict(mode="16bit", weight=t(w), bias=None if b is None else t(b))
    w = dict(embed=t("embed"), norm_w=t("norm_w"), lm_head=lin("lm_head"),
             cos_sin=layers.build_cos_sin_cache(args, dtype, "cpu", 512),
             layers=[dict(input_norm_w=t(f"l{i}.input_norm_w"), post_norm_w=t(f"l{i}.post_norm_w"),
                          qkv=lin(f"l{i}.qkv_w", f"l{i}.qkv_b"), o=lin(f"l{i}.o_w"), gate_up=lin(f"l{i}.gate_up_w"),
                          down=lin(f"l{i}.down_w")) for i in range(args.n_layers)])
    return args, w, torch.from_numpy(z["prompt"]).long(), torch.from_numpy(z["tokens"]).long(), torch.from_numpy(z["logits"])


def test_oracle_model_reproduces_hf_qwen2_fixture():
    """fp32 end to end: same greedy tokens, logits to fp32 summation-order noise; non-trivial page ids and page size 16"""
    args, w, prompt, tokens, logits = _hf_fixture()
    m = omodel.OracleQwen2(args, w, torch.float32)
    got_t, got_l = omodel.greedy_generate(m, prompt, tokens.numel(), block_size=16, block_ids=[5, 0, 3, 7])
    assert torch.equal(got_t, tokens)
    for a, b in zip(got_l, logits):
        assert ((a - b).norm() / b.norm()).item() < 2e-5


def test_oracle_model_matches_live_hf_when_available():
    """the same comparison against a transformers model built here from the fixture's weights (chunked prefill too)"""
    tr = pytest.importorskip("transformers")
    args, w, prompt, tokens, _ = _hf_fixture()
    cfg = tr.Qwen2Config(vocab_size=args.vocab_size, hidden_size=args.hidden_size, intermediate_size=args.intermediate_size,
                         num_hidden_layers=args.n_layers, num_attention_heads=args.n_heads,
                         num_key_value_heads=args.n_kv_heads, max_position_embeddings=512, rms_norm_eps=args.rms_norm_eps,
                         rope_theta=args.rope_theta, tie_word_embeddings=False, attn_implementation="eager")
    hf = tr.Qwen2ForCausalLM(cfg).eval().float()
    sd = {"model.embed_tokens.weight": w["embed"], "model.norm.weight": w["norm_w"], "lm_head.weight": w["lm_head"]["weight"]}
    q, kv, I = args.n_heads * args.head_dim, args.n_kv_heads * args.head_dim, args.intermediate_size
    for i, l in enumerate(w["layers"]):
        p = f"model.layers.{i}."
        for name, lo, hi in (("q_proj", 0, q), ("k_proj", q, q + kv), ("v_proj", q + kv, q + 2 * kv)):
            sd[p + f"self_attn.{name}.weight"] = l["qkv"]["weight"][lo:hi]
            sd[p + f"self_attn.{name}.bias"] = l["qkv"]["bias"][lo:hi]
        sd[p + "self_attn.o_proj.weight"] = l["o"]["weight"]
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = l["gate_up"]["weight"][:I], l["gate_up"]["weight"][I:]
        sd[p + "mlp.down_proj.weight"] = l["down"]["weight"]
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = l["input_norm_w"], l["post_norm_w"]
    hf.load_state_dict(sd)
    seq = torch.cat([prompt, tokens])
    with torch.no_grad():
        ref = hf(seq[None]).logits[0]                       # [L, V] all positions, causal
    m = omodel.OracleQwen2(args, w, torch.float32)
    L, bs = seq.numel(), 16
    blocks = [2, 4, 1]
    # one-shot prefill of the whole sequence
    md = orc.build_batch_metadata([L], [L], [blocks], bs)
    caches = lambda: ([torch.zeros(5, bs, args.n_kv_heads, args.head_dim) for _ in range(args.n_layers)],
                      [torch.zeros(5, bs, args.n_kv_heads, args.head_dim) for _ in range(args.n_layers)])
    kc, vc = caches()
    got = m.logits(m.forward(seq, torch.arange(L), md, kc, vc, "prefill"))
    assert ((got - ref).norm() / ref.norm()).item() < 2e-5
    # prefill 20 tokens, then the rest as ONE chunked-prefill step over the cached prefix (bottom-right causal mask)
    kc, vc = caches()
    m.forward(seq[:20], torch.arange(20), orc.build_batch_metadata([20], [20], [blocks], bs), kc, vc, "prefill")
    md = orc.build_batch_metadata([L], [L - 20], [blocks], bs)
    got = m.logits(m.forward(seq[20:], torch.arange(20, L), md, kc, vc, "chunked"))
    assert ((got - ref[20:]).norm() / ref[20:].norm()).item() < 2e-5


def _tiny(mode, dtype, seed=3):
    args = layers.ModelArgs(128, 2, 4, 2, 32, 256, 500, 1e-6, 1e4, 1024)
    hip_like = layers.Qwen2Model(args, mode, dtype, "cpu", seed=seed)
    return args, omodel.export_weights(hip_like)


@pytest.mark.parametrize("mode,dtype", [("16bit", torch.bfloat16), ("int8", torch.bfloat16), ("fp8", torch.bfloat16),
                                        ("16bit", torch.float16)])
def test_oracle_model_phases_agree(mode, dtype):
    """logits of the last token: one-shot prefill == prefill + chunked prefill == prefill + decode (to 16-bit noise;
    the quantised modes re-quantise per tensor / per token, so they get a wider, still small, band)"""
    args, w = _tiny(mode, dtype)
    m = omodel.OracleQwen2(args, w, dtype)
    g = torch.Generator().manual_seed(1)
    lens, bs = [23, 40, 7], 16
    toks = [torch.randint(0, args.vocab_size, (L,), generator=g) for L in lens]
    blocks = [[3, 9, 1], [0, 5, 8], [2]]

    def caches():
        mk = lambda: [torch.zeros(10, bs, args.n_kv_heads, args.head_dim, dtype=dtype) for _ in range(args.n_layers)]
        return mk(), mk()

    def run(cached, upto, kv, phase):
        md = orc.build_batch_metadata(upto, [u - c for u, c in zip(upto, cached)], blocks, bs)
        ids = torch.cat([t[c:u] for t, c, u in zip(toks, cached, upto)])
        pos = torch.cat([torch.arange(c, u) for c, u in zip(cached, upto)])
        h = m.forward(ids, pos, md, kv[0], kv[1], phase)
        last = md["q_cu_seq_lens"][1:].long() - 1
        return m.logits(h[last]).float()

    zero = [0] * 3
    one = run(zero, lens, caches(), "prefill")
    kv = caches()
    run(zero, [L // 2 for L in lens], kv, "prefill")
    chunk = run([L // 2 for L in lens], lens, kv, "chunked")
    kv2 = caches()
    run(zero, [L - 1 for L in lens], kv2, "prefill")
    dec = run([L - 1 for L in lens], lens, kv2, "decode")
    bar = 2e-2 if mode == "16bit" else 6e-2
    for other in (chunk, dec):
        assert ((other - one).norm() / one.norm()).item() < bar
    if mode != "fp8":   # (fp8 quantises per TENSOR, so a row's numbers depend on what else is in the batch)
        assert torch.equal(kv[0][0], kv2[0][0])     # layer-0 K cache rows are the same numbers whatever the schedule


def test_quantised_modes_track_the_dequantised_model():
    """int8: the W8A8 model's logits stay close to the 16-bit model run on the de-quantised weights (activation
    quantisation noise only); a scale applied twice or on the wrong axis would be off by orders of magnitude"""
    args, w8 = _tiny("int8", torch.bfloat16, seed=5)
    wd = dict(w8)
    wd["layers"] = []
    for l in w8["layers"]:
        l2 = dict(l)
        for k in ("qkv", "o", "gate_up", "down"):
            q = l[k]
            l2[k] = dict(mode="16bit", weight=(q["weight"].float() * q["w_scale"][:, None]).bfloat16(), bias=q["bias"])
        wd["layers"].append(l2)
    g = torch.Generator().manual_seed(2)
    toks = torch.randint(0, args.vocab_size, (33,), generator=g)
    outs = []
    for w in (w8, wd):
        m = omodel.OracleQwen2(args, w, torch.bfloat16)
        kc = [torch.zeros(3, 16, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16) for _ in range(args.n_layers)]
        vc = [torch.zeros(3, 16, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16) for _ in range(args.n_layers)]
        md = orc.build_batch_metadata([33], [33], [[2, 0, 1]], 16)
        outs.append(m.logits(m.forward(toks, torch.arange(33), md, kc, vc, "prefill")).float())
    assert ((outs[0] - outs[1]).norm() / outs[1].norm()).item() < 5e-2
