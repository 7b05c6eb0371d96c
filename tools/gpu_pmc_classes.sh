#!/bin/bash
# HBM bytes (FETCH_SIZE x2, WRITE_SIZE) per class of work of the level kernel: one split_kinds batch per counter,
# dispatches joined in order with the engine's launch trace.  usage: tools/gpu_pmc_classes.sh [PROBE_OPTS]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export PROBE_OPTS="$1"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C -d $OUT/cls_$C -o p -- python $ROOT/tools/probe_class_run.py > $OUT/cls_$C.log 2>&1 )
done
python - <<'PY'
import sqlite3, glob, re, collections
agg = collections.defaultdict(lambda: dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0, alg=0.0, ms=0.0, n=0))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    trace = [l for l in open(f"gpurun_out/cls_{c}.log") if l.startswith("[mibn launch]")]
    names = [re.match(r"\[mibn launch\] (\S+(?: \S+)*?)\s+wgs", l).group(1).strip() for l in trace]
    mbs = [float(re.search(r"MB\s+([0-9.]+)", l).group(1)) for l in trace]
    db = glob.glob(f"gpurun_out/cls_{c}/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, counter_value, duration from pmc_events where name like '%ve_level%' order by dispatch_id"))
    assert len(rows) == len(names), (len(rows), len(names))
    for (d, v, dur), n, mb in zip(rows, names, mbs):
        a = agg[n]
        a[c] += v * 1024
        if c == "FETCH_SIZE":
            a["alg"] += mb * 1e6; a["ms"] += dur / 1e6; a["n"] += 1
print(f"{'class':30s} {'launches':>8s} {'ms':>8s} {'alg GB':>8s} {'2xFETCH':>8s} {'WRITE':>8s} {'traffic/alg':>11s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    t = 2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]
    print(f"{n:30s} {a['n']:8d} {a['ms']:8.2f} {a['alg']/1e9:8.2f} {2*a['FETCH_SIZE']/1e9:8.2f} {a['WRITE_SIZE']/1e9:8.2f} {t/max(a['alg'],1):11.3f}")
PY
find $OUT -name "*.db" -size +5M -delete
