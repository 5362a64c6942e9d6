#!/bin/bash
# matrix-pipe / LDS / VALU occupancy of the flash prefill kernel by PMC (one counter group per pass; --kernel-trace only, no other trace domain):
# what the "56 % of the pipe, LDS ~70 %, VALU ~50 %" accounting of profiles/r06_prefill_attention.txt rests on
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  rm -rf /tmp/pf_pmc
  PF_N=2 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pf_pmc -- python $R/tools/prefill_attn_one.py > /tmp/pf_pmc.log 2>&1
  python - "$grp" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob("/tmp/pf_pmc/**/*.db", recursive=True)
if not dbs:
    print("[prefill pmc] no database for", sys.argv[1]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
for name in sys.argv[1].split():
    try:
        row = cur.execute("select sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? and kernel_name like '%flash_prefill_m32%'", (name,)).fetchone()
        if row and row[1]:
            print(f"[prefill pmc] {name}: {row[0] / row[1]:.4g} per dispatch ({row[1]} dispatches)")
        else:
            print(f"[prefill pmc] {name}: not collected")
    except Exception as e:
        print(f"[prefill pmc] {name}: {e!r}")
PY
done
