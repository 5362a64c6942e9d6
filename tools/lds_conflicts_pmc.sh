#!/bin/bash
# LDS bank-conflict cycles per LDS instruction for every kernel of the decode step + prefill chunk (rocprofv3 --pmc, its own run: --kernel-trace only)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lds_pmc
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/lds_pmc -- python $R/bench.py ${LDS_PMC_ARGS:---no-cpu-baseline --no-engine --no-gemm --no-pmc --no-per-rank --no-allocator-pages --steps 2 --warmup 1} > /tmp/lds_pmc.log 2>&1
python - <<'PY'
import glob, sqlite3
dbs = glob.glob("/tmp/lds_pmc/**/*.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
agg = {}
for k, c, v, n in rows:
    agg.setdefault(k, {})[c] = (v, n)
out = []
for k, d in agg.items():
    if "SQ_INSTS_LDS" not in d or d["SQ_INSTS_LDS"][0] == 0 or "xm::" not in k:
        continue
    insts, n = d["SQ_INSTS_LDS"]
    conf = d.get("SQ_LDS_BANK_CONFLICT", (0, n))[0]
    out.append((conf / insts, conf / n, insts / n, n, k))
print("# conflict cycles per LDS instruction | conflict cycles per dispatch | LDS instructions per dispatch | dispatches | kernel")
for r in sorted(out, reverse=True):
    print(f"{r[0]:8.2f} {r[1]:12.3g} {r[2]:12.3g} {r[3]:6d}  {r[4][:110]}")
PY
