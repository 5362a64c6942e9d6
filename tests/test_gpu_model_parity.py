"""LOGITS-level parity: the HIP path (xllm_amd.layers.Qwen2Model over the C ABI) against the end-to-end oracle model
(oracle/model.py, itself pinned on the HuggingFace Qwen2 implementation in tests/test_oracle_model.py) on identical weights,
tokens, page tables and caches.

The north star's bars are on logits: <= 1e-3 relative for bf16, <= 2e-2 for fp8 (BASELINE.json). "Relative" here is the
relative L2 error of a token's logits vector, ||hip - oracle|| / ||oracle||, the bar the reference's own end-to-end checks
use in spirit (its layer tests compare with rtol/atol on whole tensors). W8A8 int8 has no bar in the north star; the one
held here is stated in BARS. Greedy token ids must be equal unless the oracle's own top-2 margin is inside the error.
Every measured error is appended to gpurun_out/model_parity.jsonl so the numbers behind the asserts are on record.
"""
import json
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda"

# relative L2 error of the logits vector per generated token (max over tokens), by linear-layer mode
BARS = {"16bit": 1e-3, "int8": 5e-3, "fp8": 2e-2}


def _record(**kw):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "model_parity.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _check_token(hip_logits, orc_logits, where):
    """argmax equal, unless the oracle's top-2 margin is smaller than twice the largest logit difference"""
    if int(hip_logits.argmax()) == int(orc_logits.argmax()):
        return
    top2 = orc_logits.float().topk(2).values
    assert (top2[0] - top2[1]).item() <= 2 * (hip_logits.float() - orc_logits.float()).abs().max().item(), where


class HipSeqRunner:
    """drives layers.Qwen2Model through the product's own host builder (attention.build_batch_input) for a set of
    sequences with fixed page lists: feed(cached, upto, phase) -> logits of every sequence's last fed token"""

    def __init__(self, model, blocks, block_size, n_blocks, cache_init=None):
        from xllm_amd.attention import KVCache
        self.m, self.blocks, self.bs = model, blocks, block_size
        a = model.args
        shape = (n_blocks, block_size, model.layers[0].nkv, a.head_dim)
        if cache_init is None:
            self.caches = [KVCache(torch.zeros(shape, dtype=model.dtype, device=DEV),
                                   torch.zeros(shape, dtype=model.dtype, device=DEV)) for _ in model.layers]
        else:
            self.caches = [KVCache(k.to(DEV), v.to(DEV)) for k, v in cache_init]

    def feed(self, tokens_per_seq, cached, upto, phase):
        from xllm_amd import attention
        bi = attention.build_batch_input(cached, upto, self.blocks, self.bs)
        md = attention.build_attention_metadata(bi, phase == "prefill", phase == "chunked", DEV)
        ids = torch.cat([t[c:u] for t, c, u in zip(tokens_per_seq, cached, upto)]).to(DEV)
        hidden = self.m.forward(ids, bi.positions.long().to(DEV), md, self.caches)
        last = (bi.q_cu_seq_lens[1:].long() - 1).to(DEV)
        return self.m.logits(hidden[last]).float().cpu()


@pytest.mark.parametrize("mode", ["16bit", "int8", "fp8"])
def test_qwen2_0_5b_prefill_and_greedy_decode_logits(mode):
    """BASELINE config 1's geometry (Qwen2-0.5B: H=896, L=24, 14/2 heads, d=64, I=4864, V=151936), bs=1, prompt 128,
    10 greedy tokens; bf16 activations, linears per `mode`. The oracle generates greedily; the HIP side is fed the oracle's
    tokens (teacher forcing) so that every step compares logits on identical inputs, and its own argmax must agree."""
    from oracle import model as omodel
    from xllm_amd import layers
    args = layers.ModelArgs.qwen2_0_5b()
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=17)
    w = omodel.export_weights(model)
    om = omodel.OracleQwen2(args, w, torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    L, n_new, bs = 128, 10, 128
    prompt = torch.randint(0, args.vocab_size, (L,), generator=g)
    blocks = [3, 1]                                    # non-contiguous pages; the second one is entered at token 129
    o_tok, o_log = omodel.greedy_generate(om, prompt, n_new, bs, blocks)
    run = HipSeqRunner(model, [blocks], bs, 4)
    seq = torch.cat([prompt, o_tok])
    errs = []
    lg = run.feed([seq], [0], [L], "prefill")[0]
    for i in range(n_new):
        errs.append(_rel(lg, o_log[i]))
        _check_token(lg, o_log[i], f"step {i}")
        if i + 1 < n_new:
            lg = run.feed([seq], [L + i], [L + i + 1], "decode")[0]
    _record(test="qwen2_0_5b", mode=mode, rel_l2_per_token=errs, bar=BARS[mode])
    assert max(errs) <= BARS[mode], errs
    # the KV caches the two sides wrote agree as well (layer 0: same inputs -> bit-equal rows for 16-bit / int8 linears)
    k_hip = run.caches[0].k_cache.cpu()
    # re-run the oracle's layer-0 cache from its own generate: rebuild by a second prefill over the final sequence
    kcs = [torch.zeros(4, bs, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16) for _ in range(args.n_layers)]
    vcs = [torch.zeros(4, bs, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16) for _ in range(args.n_layers)]
    from oracle import oracle as orc
    n = L + n_new - 1
    om.forward(seq[:n], torch.arange(n), orc.build_batch_metadata([n], [n], [blocks], bs), kcs, vcs, "prefill")
    if mode != "fp8":   # fp8 quantises per tensor: prefill-in-one-go and prefill + decode see different scales
        same = (k_hip[[3, 1]].view(-1, args.n_kv_heads * args.head_dim)[:n] ==
                kcs[0][[3, 1]].view(-1, args.n_kv_heads * args.head_dim)[:n]).float().mean().item()
        assert same >= 0.999, same      # layer 0 K rows: identical up to isolated 1-ulp roundings of the qkv GEMM


@pytest.mark.parametrize("mode", ["16bit", "int8"])
def test_qwen2_7b_geometry_ragged_decode_logits(mode):
    """two Qwen2-7B-geometry layers (H=3584, 28/4 heads, d=128, I=18944), ONE decode step over a ragged batch whose caches
    hold random rows (so attention is exercised at real lengths without a long CPU prefill): logits vs the oracle"""
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import layers
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=23)
    w = omodel.export_weights(model)
    om = omodel.OracleQwen2(args, w, torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    bs = 128
    lens = [1, 129, 700, 128, 333, 5, 1024, 257]          # sequence lengths AFTER this step
    B = len(lens)
    need = [(n + bs - 1) // bs for n in lens]
    perm = torch.randperm(sum(need) + 3, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    nb = sum(need) + 3
    init = [(torch.randn(nb, bs, 4, 128, generator=g).bfloat16(), torch.randn(nb, bs, 4, 128, generator=g).bfloat16())
            for _ in range(args.n_layers)]
    toks = [torch.randint(0, args.vocab_size, (n,), generator=g) for n in lens]
    run = HipSeqRunner(model, blocks, bs, nb, cache_init=[(k.clone(), v.clone()) for k, v in init])
    hip = run.feed(toks, [n - 1 for n in lens], lens, "decode")
    md = orc.build_batch_metadata(lens, [1] * B, blocks, bs)
    kcs, vcs = [k for k, _ in init], [v for _, v in init]
    ids = torch.stack([t[-1] for t in toks])
    ref = om.logits(om.forward(ids, torch.tensor([n - 1 for n in lens]), md, kcs, vcs, "decode")).float()
    errs = [_rel(hip[b], ref[b]) for b in range(B)]
    _record(test="qwen2_7b_2layer_ragged_decode", mode=mode, rel_l2_per_seq=errs, bar=BARS[mode])
    assert max(errs) <= BARS[mode], errs
    for b in range(B):
        _check_token(hip[b], ref[b], f"seq {b}")
    # the decode step wrote this step's K rows at the slots the host builder computed: layer 0 (identical inputs on both
    # sides) holds the same cache everywhere -- bit-exact index arithmetic; the new values are the qkv GEMM's, identical up
    # to isolated 1-ulp roundings, and nothing else was touched
    k_hip, k_ref = run.caches[0].k_cache.cpu(), kcs[0]
    assert (k_hip != k_ref).sum().item() <= 0.01 * B * 4 * 128
    assert ((k_hip.float() - k_ref.float()).abs() <= 2.0 ** -7 * k_ref.float().abs() + 1e-6).all()
