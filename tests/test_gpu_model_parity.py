"""LOGITS-level parity: the HIP path (xllm_amd.layers.Qwen2Model over the C ABI) against the end-to-end oracle model
(oracle/model.py, itself pinned on the HuggingFace Qwen2 implementation in tests/test_oracle_model.py) on identical weights,
tokens, page tables and caches. See tests/_model_parity.py for the two comparisons:

 * teacher-forced per operator through the whole model: integer / copy / index work bit-exact, 16-bit float operators
   <= 1e-3 relative L2 per row, fp8 linears <= 2e-2 -- the north star's bars (BASELINE.json), asserted;
 * free-running logits and greedy tokens. The north star's "<= 1e-3 rel bf16 logits" cannot hold for ANY two evaluation orders
   of a deep 16-bit pipeline -- the oracle against ITSELF with torch's matmul summation order differs by 2e-2 on the 24-layer
   Qwen2-0.5B geometry and 5e-3 on two 7B-geometry layers (measured here as the control, every run) -- so the asserted bar is
   relative: the HIP path stays within 2x the control's drift, its greedy tokens agree with the oracle's wherever the oracle's
   own top-2 margin exceeds the drift, and (16-bit mode) it is no further from the fp32-computed truth than the oracle is.
Every measured number is appended to gpurun_out/model_parity.jsonl.
"""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda"

# teacher-forced bars, relative L2 per row: 16-bit float operators (north star: bf16 <= 1e-3), fp8 linears (<= 2e-2)
FLOAT_BAR, FP8_BAR = 1e-3, 2e-2
EXACT_OPS = ("reshape_paged_cache.k", "reshape_paged_cache.v", "rotary_embedding", "fused_add_rms_norm.residual",
             "scaled_quantize.q", "scaled_quantize.scale", "act_and_mul+quant(N1).q", "act_and_mul+quant(N1).scale")


def _record(**kw):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "model_parity.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _check_teacher_forced(errs, mode, tag):
    _record(test="teacher_forced", geometry=tag, mode=mode, errors={k: [f"{e:.3e}", f"{n:.5f}"] for k, (e, n) in errs.items()})
    for name, (e, neq) in errs.items():
        if name in EXACT_OPS or (mode == "int8" and name.startswith("linear.")):
            assert e == 0.0 and neq == 0.0, (tag, mode, name, e, neq)           # integer / copy work: bit-exact
        elif mode == "fp8" and name.startswith("linear."):
            assert e <= FP8_BAR, (tag, mode, name, e)
        else:
            assert e <= FLOAT_BAR and neq <= 0.01, (tag, mode, name, e, neq)


def _check_token(hip_logits, orc_logits, drift, where):
    """argmax equal, unless the oracle's top-2 margin is inside the evaluation-order drift of the logits"""
    if int(hip_logits.argmax()) == int(orc_logits.argmax()):
        return
    top2 = orc_logits.float().topk(2).values
    assert (top2[0] - top2[1]).item() <= 4 * drift * orc_logits.float().norm().item() / orc_logits.numel() ** 0.5, where


def _pages(lens, bs, g, spare=3):
    need = [(n + bs - 1) // bs for n in lens]
    perm = torch.randperm(sum(need) + spare, generator=g).tolist()
    blocks, used = [], 0
    for n in need:
        blocks.append(perm[used:used + n]); used += n
    return blocks, sum(need) + spare


@pytest.mark.parametrize("mode", ["16bit", "int8", "fp8"])
def test_teacher_forced_operator_parity_7b_geometry_decode(mode):
    """two Qwen2-7B-geometry layers (H=3584, 28/4 heads, d=128, I=18944), one ragged decode step over random caches: every
    operator of the step on the oracle's inputs"""
    import _model_parity as mp
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=23, fuse=False)
    trace = []
    om = omodel.OracleQwen2(args, omodel.export_weights(model), torch.bfloat16, trace=trace)
    g = torch.Generator().manual_seed(9)
    bs, lens = 128, [1, 129, 700, 128, 333, 5, 1024, 257]
    B = len(lens)
    blocks, nb = _pages(lens, bs, g)
    init = [(torch.randn(nb, bs, 4, 128, generator=g).bfloat16(), torch.randn(nb, bs, 4, 128, generator=g).bfloat16())
            for _ in range(args.n_layers)]
    ids = torch.randint(0, args.vocab_size, (B,), generator=g)
    pos = torch.tensor([n - 1 for n in lens])
    md = orc.build_batch_metadata(lens, [1] * B, blocks, bs)
    kcs, vcs = [k.clone() for k, _ in init], [v.clone() for _, v in init]
    om.forward(ids, pos, md, kcs, vcs, "decode")
    bi = attention.build_batch_input([n - 1 for n in lens], lens, blocks, bs)
    amd = attention.build_attention_metadata(bi, False, False, DEV)
    assert torch.equal(bi.new_cache_slots, md["new_cache_slots"]) and torch.equal(bi.block_tables, md["block_tables"])

    def attn_inputs(li):
        return dict(md=amd, caches=KVCache(init[li][0].to(DEV), init[li][1].to(DEV)), k_after=kcs[li], v_after=vcs[li])

    errs = mp.teacher_forced_errors(model, mp.split_trace(trace, args.n_layers), pos, "decode", attn_inputs)
    _check_teacher_forced(errs, mode, "qwen2_7b_2layer_ragged_decode")


@pytest.mark.parametrize("mode", ["16bit", "int8", "fp8"])
def test_teacher_forced_operator_parity_0_5b_prefill(mode):
    """BASELINE config 1's geometry (Qwen2-0.5B: H=896, L=24, 14/2 heads, d=64, I=4864, V=151936), all 24 layers, a varlen
    PREFILL of two sequences (128 and 57 tokens): every operator on the oracle's inputs"""
    import _model_parity as mp
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs.qwen2_0_5b()
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=17, fuse=False)
    trace = []
    om = omodel.OracleQwen2(args, omodel.export_weights(model), torch.bfloat16, trace=trace)
    g = torch.Generator().manual_seed(5)
    bs, lens = 128, [128, 57]
    blocks, nb = _pages(lens, bs, g, spare=2)
    ids = torch.randint(0, args.vocab_size, (sum(lens),), generator=g)
    pos = torch.cat([torch.arange(n) for n in lens])
    md = orc.build_batch_metadata(lens, lens, blocks, bs)
    zeros = lambda: torch.zeros(nb, bs, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16)
    kcs, vcs = [zeros() for _ in range(args.n_layers)], [zeros() for _ in range(args.n_layers)]
    om.forward(ids, pos, md, kcs, vcs, "prefill")
    bi = attention.build_batch_input([0, 0], lens, blocks, bs)
    amd = attention.build_attention_metadata(bi, True, False, DEV)

    def attn_inputs(li):
        return dict(md=amd, caches=KVCache(zeros().to(DEV), zeros().to(DEV)), k_after=kcs[li], v_after=vcs[li])

    errs = mp.teacher_forced_errors(model, mp.split_trace(trace, args.n_layers), pos, "prefill", attn_inputs)
    _check_teacher_forced(errs, mode, "qwen2_0_5b_24layer_prefill")


def _to_fp32(w):
    if isinstance(w, dict):
        return {k: _to_fp32(v) for k, v in w.items()}
    if isinstance(w, list):
        return [_to_fp32(v) for v in w]
    return w.float() if torch.is_tensor(w) and w.is_floating_point() else w


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


def _teacher_forced_logits(om, seq, L, n_new, bs, blocks):
    """logits of an oracle model for positions L-1 .. L+n_new-2 of `seq` (prefill of the prompt, then decode steps)"""
    from oracle import oracle as orc
    a = om.args
    nb = max(blocks) + 1
    kcs = [torch.zeros(nb, bs, a.n_kv_heads, a.head_dim, dtype=om.dtype) for _ in om.w["layers"]]
    vcs = [torch.zeros(nb, bs, a.n_kv_heads, a.head_dim, dtype=om.dtype) for _ in om.w["layers"]]
    out = []
    h = om.forward(seq[:L], torch.arange(L), orc.build_batch_metadata([L], [L], [blocks], bs), kcs, vcs, "prefill")
    out.append(om.logits(h[-1:]).float()[0])
    for i in range(n_new - 1):
        cur = L + i + 1
        h = om.forward(seq[cur - 1:cur], torch.tensor([cur - 1]), orc.build_batch_metadata([cur], [1], [blocks], bs), kcs, vcs,
                       "decode")
        out.append(om.logits(h).float()[0])
    return out


@pytest.mark.parametrize("mode", ["16bit", "int8", "fp8"])
def test_free_running_qwen2_0_5b_prefill_and_greedy_decode(mode):
    """BASELINE config 1's geometry, bs=1, prompt 128, 10 greedy tokens; bf16 activations, linears per `mode`. The oracle
    generates greedily; the HIP side is fed the oracle's tokens so that every step compares logits on identical inputs.
    Control: the oracle with another evaluation order of its linears on the same tokens."""
    import _model_parity as mp
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
    seq = torch.cat([prompt, o_tok])
    control = _teacher_forced_logits(mp.AltOrderOracle(args, w, torch.bfloat16), seq, L, n_new, bs, blocks)
    drift = [_rel(c, o) for c, o in zip(control, o_log)]
    run = HipSeqRunner(model, [blocks], bs, 4)
    errs = []
    lg = run.feed([seq], [0], [L], "prefill")[0]
    for i in range(n_new):
        errs.append(_rel(lg, o_log[i]))
        _check_token(lg, o_log[i], max(max(drift), errs[-1]), f"step {i}")
        if i + 1 < n_new:
            lg = run.feed([seq], [L + i], [L + i + 1], "decode")[0]
    rec = dict(test="free_running_qwen2_0_5b", mode=mode, hip_vs_oracle=errs, control_oracle_alt_order_vs_oracle=drift)
    if mode == "16bit":   # distance to the fp32-computed truth (same bf16-valued weights, fp32 activations and arithmetic)
        w32 = _to_fp32(w)
        truth = _teacher_forced_logits(omodel.OracleQwen2(args, w32, torch.float32), seq, L, n_new, bs, blocks)
        e_orc = [_rel(o, t) for o, t in zip(o_log, truth)]
        lg_all = [run_l for run_l in _hip_logits_again(model, seq, L, n_new, bs, blocks)]
        e_hip = [_rel(h, t) for h, t in zip(lg_all, truth)]
        rec.update(oracle_vs_fp32_truth=e_orc, hip_vs_fp32_truth=e_hip)
        assert max(e_hip) <= 1.25 * max(e_orc), (e_hip, e_orc)
    _record(**rec)
    assert max(errs) <= 2.0 * max(drift), (errs, drift)


def _hip_logits_again(model, seq, L, n_new, bs, blocks):
    run = HipSeqRunner(model, [blocks], bs, 4)
    out = [run.feed([seq], [0], [L], "prefill")[0]]
    for i in range(n_new - 1):
        out.append(run.feed([seq], [L + i], [L + i + 1], "decode")[0])
    return out


@pytest.mark.parametrize("mode", ["16bit", "int8"])
def test_free_running_qwen2_7b_geometry_ragged_decode(mode):
    """two Qwen2-7B-geometry layers, ONE decode step over a ragged batch whose caches hold random rows (so attention is
    exercised at real lengths without a long CPU prefill): logits vs the oracle, fused (N1) and reference operator order"""
    import _model_parity as mp
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import layers
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, mode, torch.bfloat16, DEV, seed=23)
    w = omodel.export_weights(model)
    g = torch.Generator().manual_seed(9)
    bs, lens = 128, [1, 129, 700, 128, 333, 5, 1024, 257]          # sequence lengths AFTER this step
    B = len(lens)
    blocks, nb = _pages(lens, bs, g)
    init = [(torch.randn(nb, bs, 4, 128, generator=g).bfloat16(), torch.randn(nb, bs, 4, 128, generator=g).bfloat16())
            for _ in range(args.n_layers)]
    toks = [torch.randint(0, args.vocab_size, (n,), generator=g) for n in lens]
    run = HipSeqRunner(model, blocks, bs, nb, cache_init=[(k.clone(), v.clone()) for k, v in init])
    hip = run.feed(toks, [n - 1 for n in lens], lens, "decode")
    md = orc.build_batch_metadata(lens, [1] * B, blocks, bs)
    ids = torch.stack([t[-1] for t in toks])
    pos = torch.tensor([n - 1 for n in lens])
    outs = []
    for cls in (omodel.OracleQwen2, mp.AltOrderOracle):
        om = cls(args, w, torch.bfloat16)
        kcs, vcs = [k.clone() for k, _ in init], [v.clone() for _, v in init]
        outs.append((om.logits(om.forward(ids, pos, md, kcs, vcs, "decode")).float(), kcs))
    ref, kcs = outs[0]
    errs = [_rel(hip[b], ref[b]) for b in range(B)]
    drift = [_rel(outs[1][0][b], ref[b]) for b in range(B)]
    _record(test="free_running_qwen2_7b_2layer_ragged_decode", mode=mode, hip_vs_oracle=errs,
            control_oracle_alt_order_vs_oracle=drift)
    # 16-bit: a smooth drift, the HIP path within 2x the control. int8: the drift is DISCRETE (a row is either bit-close or one
    # re-rolled int8 code and its cascade away) and this control -- re-ordered LINEARS, exact in int8 -- perturbs nothing that
    # feeds a quantiser (round-4 review, weak #1): the int8 bar lives in test_free_running_int8_jumps_are_attributed_and_match_
    # the_control below, on 64 rows against controls that perturb the attention's summation order; here only the absolute
    # size of one jump is held
    if mode == "16bit":
        assert max(errs) <= 2.0 * max(drift), (errs, drift)
    else:
        assert max(errs) <= 5e-2, errs
    for b in range(B):
        _check_token(hip[b], ref[b], max(max(drift), errs[b]), f"seq {b}")
    # the decode step wrote this step's K rows at the slots the host builder computed: layer 0 (identical inputs on both
    # sides) holds the same cache everywhere -- bit-exact index arithmetic; the new values are the qkv GEMM's, identical up
    # to isolated 1-ulp roundings, and nothing else was touched
    k_hip, k_ref = run.caches[0].k_cache.cpu(), kcs[0]
    assert (k_hip != k_ref).sum().item() <= 0.01 * B * 4 * 128
    assert ((k_hip.float() - k_ref.float()).abs() <= 2.0 ** -7 * k_ref.float().abs() + 1e-6).all()


@pytest.mark.timeout(1500)
def test_free_running_int8_jumps_are_attributed_and_match_the_control():
    """round-4 review (weak #1, next #2): in a W8A8 model a free-running row is either bit-close to the oracle or a discrete
    JUMP away (1e-3 .. 4e-2 in the logits). This test shows where the jumps come from and holds their size AND frequency to a
    control that perturbs what actually feeds a quantiser -- the attention's fp32 summation order:

      * 2 weight seeds x 32 sequences (two Qwen2-7B-geometry layers, one ragged decode step over random caches);
      * controls = the oracle with its decode attention evaluated in other, equally valid, orders: keys split in 1 / 2 / 4 / 8
        ranges merged in fp32, flash-decoding's online order over 16- / 64-key tiles, softmax in base e or base 2 -- and the
        online order with P entering the PV product as hi + lo 16-bit parts (p to ~2^-17), the cast point of
        attention_decode.hip:289-300 (the matrix core takes 16-bit operands);
      * attribution, row by row (tests/_model_parity.py::attribute_rows): the first operator whose output differs from the
        oracle's and the first per-token quantiser whose int8 codes differ.

    Asserted: (a) the fused model is bit-equal to the reference operator order; (b) every HIP row is either clean (<= 3e-4: the
    16-bit lm_head's summation order) or a jump whose FIRST difference is a handful of 1-ulp elements of an attention output and
    whose first code difference is <= 4 codes of the o_proj operand moving by exactly 1 with the row's scale unchanged;
    (c) size: HIP's largest jump <= 1.25 x the controls' largest; (d) frequency: HIP's jumps <= 1.5 x the hi + lo control's + 2
    over the same 64 rows, and per differing attention element HIP jumps no more often than 1.5 x the pure-fp32 controls
    (measured: 2.9 % against 4.9 %; HIP differs from the oracle on 0.17 % of the attention elements, the fp32 re-orderings on
    0.03 %, the hi + lo control on 0.19 %)."""
    import _model_parity as mp
    tot = dict(hip_jumps=0, hilo_jumps=0, hip_diff=0, fp32_jumps=0, fp32_diff=0, rows=0)
    hip_max = ctrl_max = 0.0
    for seed in (23, 24):
        r = mp.int8_jump_experiment(seed=seed)
        B = len(r["lens"])
        assert r["fused_equals_unfused"]                                                             # (a)
        e, at = r["errors"]["hip"], r["attribution"]["hip"]
        jumps = [i for i in range(B) if e[i] > mp.JUMP]
        for i in range(B):                                                                           # (b)
            if i in jumps:
                fd, fc = at[i]["first_diff"], at[i]["first_code_diff"]
                assert fd is not None and fd[1] == "attn" and fd[2] <= 0.01 * 3584 and fd[3] <= 1e-3, (seed, i, at[i])
                assert fc is not None and fc[1] == "attn" and fc[2] <= 4 and fc[3] == 1 and not fc[4], (seed, i, at[i])
            else:
                assert e[i] <= 3e-4, (seed, i, e[i])
        names = [c[0] for c in mp.CONTROLS]
        fp32 = [n for n in names if "hilo" not in n]
        hilo = [n for n in names if "hilo" in n]
        count = lambda n: sum(1 for x in r["errors"][n] if x > mp.JUMP)
        hip_max = max(hip_max, max(e))
        ctrl_max = max([ctrl_max] + [max(r["errors"][n]) for n in names])
        tot["rows"] += B
        tot["hip_jumps"] += len(jumps)
        tot["hilo_jumps"] += max(count(n) for n in hilo)
        tot["hip_diff"] += r["attn0_diff_elements"]["hip"]
        tot["fp32_jumps"] += sum(count(n) for n in fp32)
        tot["fp32_diff"] += sum(r["attn0_diff_elements"][n] for n in fp32)
        _record(test="free_running_int8_jump_attribution", seed=seed, lens=r["lens"], errors=r["errors"],
                attn0_diff_elements=r["attn0_diff_elements"],
                hip_attribution=[dict(row=i, err=e[i], **at[i]) for i in jumps],
                control_jump_rows={n: [i for i in range(B) if r["errors"][n][i] > mp.JUMP] for n in names})
    _record(test="free_running_int8_jump_summary", hip_max=hip_max, control_max=ctrl_max, **tot)
    assert hip_max <= 1.25 * ctrl_max, (hip_max, ctrl_max)                                          # (c)
    assert tot["hip_jumps"] <= 1.5 * tot["hilo_jumps"] + 2, tot                                      # (d)
    assert tot["hip_diff"] <= 0.005 * tot["rows"] * 3584, tot
    assert tot["fp32_diff"] > 0 and tot["hip_jumps"] / tot["hip_diff"] <= 1.5 * tot["fp32_jumps"] / tot["fp32_diff"], tot


@pytest.mark.timeout(1500)
def test_teacher_forced_prefill_at_ctx4096_7b_geometry():
    """round-2 review, weak #1: the teacher-forced prefill test ran 128- and 57-token prompts; this one runs BASELINE's context
    length -- ONE sequence of 4096 tokens through two Qwen2-7B-geometry layers (16-bit mode) -- with the oracle's attention in
    the flash cast point (p_round="flash": un-normalised 64-key-tile P rounded to bf16), so the default prefill kernel is held
    to the ABSOLUTE 1e-3 per row at ctx = 4096, next to every other operator of the prefill step."""
    import _model_parity as mp
    from oracle import model as omodel
    from oracle import oracle as orc
    from xllm_amd import attention, layers
    from xllm_amd.attention import KVCache
    args = layers.ModelArgs(3584, 2, 28, 4, 128, 18944, 32000, 1e-6, 1e6, 8192)
    model = layers.Qwen2Model(args, "16bit", torch.bfloat16, DEV, seed=29, fuse=False)
    trace = []
    om = omodel.OracleQwen2(args, omodel.export_weights(model), torch.bfloat16, p_round="flash", trace=trace)
    g = torch.Generator().manual_seed(13)
    bs, lens = 128, [4096]
    blocks, nb = _pages(lens, bs, g, spare=2)
    ids = torch.randint(0, args.vocab_size, (sum(lens),), generator=g)
    pos = torch.cat([torch.arange(n) for n in lens])
    md = orc.build_batch_metadata(lens, lens, blocks, bs)
    zeros = lambda: torch.zeros(nb, bs, args.n_kv_heads, args.head_dim, dtype=torch.bfloat16)
    kcs, vcs = [zeros() for _ in range(args.n_layers)], [zeros() for _ in range(args.n_layers)]
    om.forward(ids, pos, md, kcs, vcs, "prefill")
    bi = attention.build_batch_input([0], lens, blocks, bs)
    amd = attention.build_attention_metadata(bi, True, False, DEV)

    def attn_inputs(li):
        return dict(md=amd, caches=KVCache(zeros().to(DEV), zeros().to(DEV)), k_after=kcs[li], v_after=vcs[li])

    errs = mp.teacher_forced_errors(model, mp.split_trace(trace, args.n_layers), pos, "prefill", attn_inputs)
    _check_teacher_forced(errs, "16bit", "qwen2_7b_2layer_prefill_ctx4096_flash_oracle")
    assert errs["attention.prefill"][0] <= FLOAT_BAR
