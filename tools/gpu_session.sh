#!/bin/bash
# One parametrised GPU-box session (replaces the per-session gpu_r0*_*.sh scripts of rounds 2-4; their logs stay under profiles/,
# the diary in profiles/NOTES*.md).   usage: tools/gpu_session.sh <tag> <stage> [<stage> ...]
# Stages (run in the order given; every stage writes gpurun_out/<tag>_*):
#   host          nproc, cgroup CPU quota, CPU model, VRAM
#   tests[:expr]  pytest -m gpu (optionally -k expr)
#   bench         the bench line as the driver runs it (--steps 20 --warmup 5)
#   full          bench.py --full-stream (configs 3 / 4 as written: the first 1 M requests once)
#   profile       rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the bench command (tools/gpu_profile.sh)
#   sq            one pass of SQ counters over the bench command (fractions of SQ_WAVE_CYCLES per kernel)
#   calib         PMC calibration on a known byte count (tools/ubench/sweep_real)
#   threads       the few-planning-threads table (1 2 4 8 and the whole quota; THREADS="1 2" to choose)
#   ab            interleaved A/B of bench argument sets: AB="--opt x=0;--opt x=1" REPS=2 (LIBS="a.so;b.so" to A/B builds)
#   classes       tools/probe_classes.py at full launch size: one launch per (level, class of work), GB/s per class (PROBE_N=52429)
#   parity        the GPU parity suite under non-default engine options: PARITY_OPTS="mfma_kernel=1" [PARITY_K="<pytest -k expr>"]
#   planner       tools/bench_planner.py (host planning rate, one thread)
#   smoke         __graft_entry__.smoke()
#   plantrace     rocprofv3 kernel trace of a two-planning-threads rank (the device planner beside the VE kernels): calls / total / average per kernel
#   planlanes     the device planner alone (--sync, whole chunks on the device) at PLAN_LANES="1 4 16 32 64" requests per wave: kernel ms per chunk
#   plansq        SQ counters of the device planner's kernels (one pass, --sync, whole chunks on the device; PLAN_OPTS="--opt wave_plan=1")
TAG=${1:?tag}; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp

line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['pipeline_clocks_ms_per_step']; s = r.get('per_kernel_serialised', {})
        print('%-52s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned %.0f  all kernels %.0f GB/s  ' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config'].get('device_planned_requests_per_step', 0), r['all_kernels_GBps']) + '  '.join('%s %.0f' % (k.replace('ve_', '').replace('_kernel', ''), v.get('achieved', 0)) for k, v in s.items() if isinstance(v, dict)))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }

for STAGE in "$@"; do
case $STAGE in
host)
  (nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)"; rocm-smi --showmeminfo vram 2>/dev/null | grep Total) > $OUT/${TAG}_host.txt 2>&1 ;;
tests*)
  K=${STAGE#tests}; K=${K#:}
  if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -k "$K" > $OUT/${TAG}_pytest_gpu.log 2>&1
  else timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1; fi
  echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log; tail -14 $OUT/${TAG}_pytest_gpu.log | cut -c1-240 ;;
bench)
  timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
  echo "bench rc $?"; line "driver-shaped" < $OUT/${TAG}_bench.log; tail -3 $OUT/${TAG}_bench.err | cut -c1-300 ;;
full)
  timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu --no-configs --full-stream > $OUT/${TAG}_full_stream.log 2>&1
  grep -o '"full_stream": {[^}]*}' $OUT/${TAG}_full_stream.log | head -2 ;;
profile)
  bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1; tail -24 $OUT/${TAG}_profile_session.log | cut -c1-300 ;;
sq)
  ( cd /tmp; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/${TAG}_sq -o sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-adaptive > $OUT/${TAG}_sq.log 2>&1 )
  python3 - "$(find $OUT/${TAG}_sq -name '*.db' | head -1)" <<'PY' | tee $OUT/${TAG}_sq_counters.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
by = {}
for n, c, k, v in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
    if "mibn" in n: by.setdefault(n.split("(")[0].split("::")[-1], {})[c] = (k, v)
for n, d in by.items():
    wc = d.get("SQ_WAVE_CYCLES", (0, 1))[1] or 1
    print("%-24s launches %5d  " % (n[:24], d.get("SQ_WAVE_CYCLES", (0, 0))[0]) + "  ".join("%s %.3f" % (c.replace("SQ_", ""), v / wc) for c, (k, v) in sorted(d.items()) if c != "SQ_WAVE_CYCLES") + "  (fractions of SQ_WAVE_CYCLES)")
PY
  find $OUT -name "*.db" -delete ;;
calib)
  [ -x tools/ubench/sweep_real ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/sweep_real tools/ubench/sweep_real.hip
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 200 rocprofv3 --pmc $c -d $OUT/${TAG}_calib_$c -o c -- $ROOT/tools/ubench/sweep_real 1024 8 > $OUT/${TAG}_calib_$c.log 2>&1 )
    python3 - "$(find $OUT/${TAG}_calib_$c -name '*.db' | head -1)" $c <<'PY' | tee -a $OUT/${TAG}_pmc_calibration.log
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, k, v in cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name"):
    if "sweep" in n: print("%s %s launches %d avg %.1f KB = %.3f GB per launch; known: 1024 requests x 8 MiB = %.3f GB each way" % (c, n[:48], k, v, v * 1024 / 1e9, 1024 * 8 * 1048576 / 1e9))
PY
  done
  find $OUT -name "*.db" -delete ;;
threads)
  for t in ${THREADS:-1 2 4 8 0}; do
    a="--threads $t"; [ "$t" = 0 ] && a=""
    timeout 300 python bench.py --steps ${STEPS:-4} --warmup ${WARMUP:-3} --no-cpu --no-configs $a 2>&1 | line "${a:-whole quota}" | tee -a $OUT/${TAG}_threads.log
  done ;;
ab)
  IFS=';' read -ra SETS <<< "${AB:-}"; IFS=';' read -ra LIBSET <<< "${LIBS:-libmibn.so}"
  [ ${#SETS[@]} = 0 ] && SETS=("")
  for rep in $(seq 1 ${REPS:-2}); do for lib in "${LIBSET[@]}"; do for args in "${SETS[@]}"; do
    MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu --no-configs ${ADAPT:---no-adaptive} $args 2>&1 | line "$lib $args" | tee -a $OUT/${TAG}_ab.log
  done; done; done ;;
classes)
  PROBE_N=${PROBE_N:-52429} PROBE_OPTS=${PROBE_OPTS:-arena_gb=250} timeout 600 python tools/probe_classes.py 2>&1 | tee $OUT/${TAG}_probe_classes.log | tail -40 ;;
parity)
  MIBN_OPTS="${PARITY_OPTS:?}" timeout 1500 python -m pytest tests -m gpu -x -q -k "${PARITY_K:-golden or stream or c3 or grid or stratified or heavy or sweep or device_planner}" > $OUT/${TAG}_pytest_parity.log 2>&1
  echo "pytest rc $? (MIBN_OPTS=$PARITY_OPTS)" >> $OUT/${TAG}_pytest_parity.log; tail -6 $OUT/${TAG}_pytest_parity.log | cut -c1-240 ;;
planner)
  timeout 600 python tools/bench_planner.py 2>&1 | tee $OUT/${TAG}_planner.log | tail -12 ;;
planlanes)
  for l in ${PLAN_LANES:-1 4 16 32 64}; do
    timeout 300 python bench.py --steps 2 --warmup 1 --global-batch 32768 --batch 32768 --sync --no-cpu --no-configs --no-adaptive --opt gpu_emit=1 --opt emit_share=1 --opt trace=1 --opt plan_lanes=$l ${PLAN_OPTS:-} 2>&1 | grep "mibn plan\] chunk" | tail -2 | sed "s/^/lanes $l ${PLAN_OPTS:-}: /" | tee -a $OUT/${TAG}_planlanes.log
  done ;;
plansq)
  ( cd /tmp; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/${TAG}_plansq -o sq -- python $ROOT/bench.py --steps 2 --warmup 1 --global-batch 32768 --batch 32768 --sync --no-cpu --no-configs --no-adaptive --opt gpu_emit=1 --opt emit_share=1 ${PLAN_OPTS:-} > $OUT/${TAG}_plansq.log 2>&1 )
  python3 - "$(find $OUT/${TAG}_plansq -name '*.db' | head -1)" <<'PY' | tee $OUT/${TAG}_plansq_counters.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
by = {}
for n, c, k, v in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
    if "order_kernel" in n or "emit_kernel" in n or "wave_plan" in n: by.setdefault(n.split("(")[0].split("::")[-1], {})[c] = (k, v)
for n, d in by.items():
    wc = d.get("SQ_WAVE_CYCLES", (0, 1))[1] or 1
    print("%-24s launches %5d  " % (n[:24], d.get("SQ_WAVE_CYCLES", (0, 0))[0]) + "  ".join("%s %.3f" % (c.replace("SQ_", ""), v / wc) for c, (k, v) in sorted(d.items()) if c != "SQ_WAVE_CYCLES") + "  (fractions of SQ_WAVE_CYCLES)  wave cycles %.3g" % wc)
PY
  find $OUT -name "*.db" -delete ;;
plantrace)
  ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_plantrace -o t -- python $ROOT/bench.py --steps 3 --warmup 3 --no-cpu --no-configs --threads 2 > $OUT/${TAG}_plantrace.log 2>&1 )
  python3 - "$(find $OUT/${TAG}_plantrace -name '*.db' | head -1)" <<'PY' | tee $OUT/${TAG}_plantrace_summary.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) / 1e6, avg(d.end - d.start) / 1e6, max(d.end - d.start) / 1e6 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("bench.py --threads 2 (device planner on): kernel-trace, durations in ms")
for n, c, t, a, m in rows[:12]:
    print("%-60s calls %6d  total %10.2f  avg %8.3f  max %8.3f  %5.1f %%" % (n[:60], c, t, a, m, 100 * t / tot))
PY
  grep -h '"metric"' $OUT/${TAG}_plantrace.log | head -1 | cut -c1-400 >> $OUT/${TAG}_plantrace_summary.txt
  find $OUT -name "*.db" -delete ;;
smoke)
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/${TAG}_smoke.log ;;
*) echo "unknown stage $STAGE" ;;
esac
done
find $OUT -name "*.db" -size +20M -delete
