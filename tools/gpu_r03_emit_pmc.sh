#!/bin/bash
# round 3: HBM traffic of the device planner's kernels (PMC passes, kernels serialised by the counter collection)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-adaptive --threads 2 --opt gpu_emit=1 --opt emit_share=1 $EXTRA"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/emit_pmc_$c -o pmc -- $CMD > $OUT/emit_pmc_$c.log 2>&1
done
cd $ROOT
python - <<'PY'
import sqlite3, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    db = glob.glob('gpurun_out/emit_pmc_%s/**/*.db' % c, recursive=True)
    if not db: print('no db for', c); continue
    con = sqlite3.connect(db[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pc = [t for t in tabs if 'pmc_event' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    q = f"select s.kernel_name, count(*), sum(p.value), avg(d.end - d.start) from {pc} p join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name"
    try:
        for name, n, v, dur in con.execute(q):
            print('%-12s %-40s launches %4d  total %10.1f MB (counter units of KB)  avg kernel %.2f ms' % (c, name[:40], n, v / 1e3, dur / 1e6))
    except Exception as e:
        print('query failed', e, tabs)
PY
find $OUT -name "*.db" -size +20M -delete
