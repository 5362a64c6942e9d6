"""The committed parity records are what a run of the GPU tests produced: hold them to the SAME bars on the CPU.

tests/test_gpu_model_parity.py appends every logits-level parity record of a run to gpurun_out/model_parity.jsonl; the copy under
profiles/ is evidence the design document cites. Nothing else ties that copy to the bars the tests assert (round-2 review, weak
#2), so this test re-applies them to the committed file: a record that no longer meets a bar (a stale copy, an edited number, a
loosened test) fails here, without a GPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import glob as _glob
RECORDS = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r0*_model_parity.jsonl")))[-1]    # the latest round's records

# the bars of tests/test_gpu_model_parity.py (kept literal here on purpose: changing a bar has to be done in two places)
FLOAT_BAR, FP8_BAR = 1e-3, 2e-2
EXACT_OPS = ("reshape_paged_cache.k", "reshape_paged_cache.v", "rotary_embedding", "fused_add_rms_norm.residual",
             "scaled_quantize.q", "scaled_quantize.scale", "act_and_mul+quant(N1).q", "act_and_mul+quant(N1).scale")


def _load():
    with open(RECORDS) as f:
        return [json.loads(line) for line in f if line.strip()]


def test_committed_teacher_forced_records_meet_the_bars():
    recs = [r for r in _load() if r["test"] == "teacher_forced"]
    seen = {(r["geometry"], r["mode"]) for r in recs}
    for geo in ("qwen2_7b_2layer_ragged_decode", "qwen2_0_5b_24layer_prefill"):
        for mode in ("16bit", "int8", "fp8"):
            assert (geo, mode) in seen, f"no committed record for {geo} / {mode}"
    assert ("qwen2_7b_2layer_prefill_ctx4096_flash_oracle", "16bit") in seen      # round 3: BASELINE's context length
    for r in recs:
        for name, (e, neq) in r["errors"].items():
            e, neq = float(e), float(neq)
            if name in EXACT_OPS or (r["mode"] == "int8" and name.startswith("linear.")):
                assert e == 0.0 and neq == 0.0, (r["geometry"], r["mode"], name, e, neq)
            elif r["mode"] == "fp8" and name.startswith("linear."):
                assert e <= FP8_BAR, (r["geometry"], r["mode"], name, e)
            else:
                assert e <= FLOAT_BAR and neq <= 0.01, (r["geometry"], r["mode"], name, e, neq)


def test_committed_free_running_records_stay_inside_their_controls():
    recs = [r for r in _load() if r["test"].startswith("free_running") and "hip_vs_oracle" in r]
    assert {r["mode"] for r in recs if r["test"] == "free_running_qwen2_0_5b"} == {"16bit", "int8", "fp8"}
    for r in recs:
        errs, drift = r["hip_vs_oracle"], r["control_oracle_alt_order_vs_oracle"]
        assert len(errs) == len(drift) and len(errs) > 0
        if r["test"] == "free_running_qwen2_0_5b" or r["mode"] == "16bit":
            assert max(errs) <= 2.0 * max(drift), (r["test"], r["mode"], max(errs), max(drift))
        else:   # int8, 7B geometry: one jump's size here; frequency and attribution in the records checked below
            assert max(errs) <= 5e-2, (r["test"], r["mode"], errs)
        if "hip_vs_fp32_truth" in r:
            assert max(r["hip_vs_fp32_truth"]) <= 1.25 * max(r["oracle_vs_fp32_truth"])


def test_committed_int8_jump_records_show_the_control_jumping_like_the_hip_path():
    """round-4 review, next #2: the committed record must show (i) every HIP jump attributed to <= 4 codes of an attention-output
    quantiser moving by 1, (ii) control rows with the same discrete jumps: size within 1.25x, frequency within 1.5x + 2 of the
    hi + lo control, and per differing attention element no more often than 1.5x the pure-fp32 controls"""
    per_seed = [r for r in _load() if r["test"] == "free_running_int8_jump_attribution"]
    summ = [r for r in _load() if r["test"] == "free_running_int8_jump_summary"]
    assert len(per_seed) >= 2 and len(summ) >= 1
    for r in per_seed:
        assert any("hilo" in k for k in r["errors"]) and len(r["errors"]["hip"]) >= 32
        jumps = [i for i, e in enumerate(r["errors"]["hip"]) if e > 5e-4]
        assert [a["row"] for a in r["hip_attribution"]] == jumps
        for a in r["hip_attribution"]:
            assert a["first_diff"][1] == "attn" and a["first_code_diff"][1] == "attn"
            assert a["first_code_diff"][2] <= 4 and a["first_code_diff"][3] == 1 and not a["first_code_diff"][4]
        assert sum(len(v) for v in r["control_jump_rows"].values()) > 0      # the controls DO jump
    s = summ[-1]
    assert s["hip_max"] <= 1.25 * s["control_max"]
    assert s["hip_jumps"] <= 1.5 * s["hilo_jumps"] + 2
    assert s["hip_jumps"] / s["hip_diff"] <= 1.5 * s["fp32_jumps"] / s["fp32_diff"]


def test_full_gpu_suite_record_is_for_this_tree():
    """round-3 review (next #1d): the round's last GPU run is the FULL `pytest -m gpu` on the tree that is submitted. Its log
    (the latest profiles/rNN_pytest_gpu.txt, written by tools/final_gpu_suite.sh) starts with the digest of the sources it ran on
    (kernels, C ABI, host mirror, shim, oracle, tests: tools/source_digest.py); this test fails when any of them changed since,
    when the run was not green, or when it ran fewer tests than the suite holds."""
    import re
    import subprocess
    import sys
    recs = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r0*_pytest_gpu.txt")))
    assert recs, "no committed record of the full GPU suite"
    rec = recs[-1]                                                  # the latest round's
    text = open(rec).read()
    m = re.search(r"^# source-digest: ([0-9a-f]{64})$", text, flags=re.M)
    assert m, "the record does not carry a source digest"
    now = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "source_digest.py")]).decode().strip()
    assert m.group(1) == now, ("kernels / host code / tests changed after the last full GPU run: re-run tools/final_gpu_suite.sh "
                               "on the GPU box and commit its log as profiles/rNN_pytest_gpu.txt")
    assert re.search(r"^# pytest rc=0$", text, flags=re.M), "the recorded GPU run was not green"
    res = re.search(r"(\d+) passed", text)
    assert res and int(res.group(1)) >= 400 and " failed" not in text.split("# pytest rc")[0].splitlines()[-1]
