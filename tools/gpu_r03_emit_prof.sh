#!/bin/bash
# round 3: durations of the device planner's kernels (order_kernel, emit_kernel) beside the VE kernels (bench, async calls) and
# alone (a blocking call of one chunk): rocprofv3 kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-adaptive --threads 2 --opt gpu_emit=1 --opt emit_share=1 $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/emit_trace -o trace -- $CMD > $OUT/emit_trace.log 2>&1
cd $ROOT
T=$(find $OUT/emit_trace -name "*.db" | head -1)
python tools/rocprof_summary.py $T > $OUT/r03_emit_rocprofv3_summary.txt 2>&1
grep -h '"metric"' $OUT/emit_trace.log | head -1 | cut -c1-400 >> $OUT/r03_emit_rocprofv3_summary.txt
head -12 $OUT/r03_emit_rocprofv3_summary.txt
python - <<'PY'
import sqlite3, glob, sys
db = glob.glob('gpurun_out/emit_trace/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
t0 = rows[0][1]
for name, a, b in rows:
    if 'order_kernel' in name or 'emit_kernel' in name:
        print('%-14s start %9.2f ms  dur %8.2f ms' % (name.split('(')[0][-14:], (a - t0) / 1e6, (b - a) / 1e6))
PY
find $OUT -name "*.db" -size +20M -delete
