#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the level kernel for two engine settings (separate passes): tools/gpu_pmc_ab.sh "chain=0" "chain=1"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for OPT in "$@"; do
  i=$((i+1))
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C -d $OUT/ab_${i}_$C -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --opt $OPT > $OUT/ab_${i}_$C.log 2>&1 )
  done
done
python - "$@" <<'PY'
import sqlite3, glob, sys, json
for i, opt in enumerate(sys.argv[1:], 1):
    row = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for db in glob.glob(f"gpurun_out/ab_{i}_{c}/**/*.db", recursive=True):
            cur = sqlite3.connect(db).cursor()
            tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
            pm = [t for t in tabs if t.startswith("rocpd_pmc_event")] 
            try:
                rows = list(cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"))
            except Exception as e:
                rows = []
                print("no pmc_events view:", e, tabs[:8])
            for n, cn, k, v in rows:
                if "ve_level" in n:
                    row[cn] = (k, v)
        line = [l for l in open(f"gpurun_out/ab_{i}_{c}.log") if l.startswith('{"metric"')]
        if line:
            d = json.loads(line[-1]); row["alg"] = d["roofline"]["alg_bytes_per_launch"] * d["roofline"]["launches"]
    f = row.get("FETCH_SIZE", (0, 0)); w = row.get("WRITE_SIZE", (0, 0))
    print(f"{opt:24s} launches {f[0]} fetch {2*f[1]*1024/1e9:8.2f} GB (x2 corrected) write {w[1]*1024/1e9:8.2f} GB alg {row.get('alg',0)/1e9:8.2f} GB  traffic/alg {(2*f[1]+w[1])*1024/max(row.get('alg',1),1):.3f}")
PY
find $OUT -name "*.db" -size +5M -delete
