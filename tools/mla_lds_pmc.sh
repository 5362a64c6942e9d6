#!/bin/bash
# which LDS stall counters the MLA decode kernel raises (rocprofv3 --pmc, one group per pass, tools/mla_pmc_probe.py)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LDS_IDX_ACTIVE SQ_LDS_MEM_VIOLATIONS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL" "SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/mla_pmc
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/mla_pmc -- python $R/tools/mla_pmc_probe.py 128 8192 16 4 > /tmp/mla_pmc.log 2>&1
  python - "$grp" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob("/tmp/mla_pmc/**/*.db", recursive=True)
if not dbs:
    print("[mla pmc] no database for", sys.argv[1]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
for name in sys.argv[1].split():
    row = cur.execute("select sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? and kernel_name like '%mla_decode_dma%'", (name,)).fetchone()
    print(f"[mla pmc] {name}: " + (f"{row[0] / row[1]:.4g} per dispatch ({row[1]} dispatches)" if row and row[1] else "not collected"))
PY
done
