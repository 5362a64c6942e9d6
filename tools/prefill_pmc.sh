#!/bin/bash
# HBM / fabric bytes per launch of the flash prefill kernel with and without the XCD-pinned block mapping (tuning library):
# rocprofv3 --pmc FETCH_SIZE (one counter per pass, MI355X_MICROARCH.md) around tools/prefill_attn_one.py
R="${GRAFT_REPO_ROOT:-/root/repo}"
export XLLM_MI355_LIB=$R/xllm_amd/lib/libxllm_mi355_tuning.so
cd /tmp && export TMPDIR=/tmp
for xcd in 1 0; do
  rm -rf /tmp/pf_pmc
  XLLM_MI355_PREFILL_XCD=$xcd PF_N=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_pmc -- python $R/tools/prefill_attn_one.py > /dev/null 2>&1
  python - <<PY
import glob, sqlite3
db = glob.glob("/tmp/pf_pmc/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
row = cur.execute("select sum(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like '%flash_prefill_m32%'").fetchone()
kib = row[0] / row[1]
print(f"[prefill pmc] XLLM_MI355_PREFILL_XCD=$xcd: FETCH_SIZE {kib:.0f} KiB per dispatch x 2 (gfx950 correction) = {2 * kib * 1024 / 1e6:.1f} MB of fabric reads per launch ({row[1]} dispatches)")
PY
done
