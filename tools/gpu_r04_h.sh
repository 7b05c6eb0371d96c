#!/bin/bash
# Round-4 profile session: GPU parity suite, the bench line as the driver runs it, the whole 1 M-request stream, rocprofv3 trace + PMC
# passes of the bench command, PMC calibration on a known byte count, the few-host-threads scenario.  usage: tools/gpu_r04_h.sh <tag>
TAG=${1:-r04_h}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)"; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > $OUT/${TAG}_host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -22 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
echo "bench rc $?"; tail -c 1200 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu --no-configs --full-stream > $OUT/${TAG}_full_stream.log 2>&1
grep -o '"full_stream": {[^}]*}' $OUT/${TAG}_full_stream.log | head -2
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
tail -30 $OUT/${TAG}_profile_session.log
# where the waves' cycles go (one pass of SQ counters over the same command; quad-cycle units, see MI355X_MICROARCH.md)
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/${TAG}_sq -o sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-adaptive > $OUT/${TAG}_sq.log 2>&1
DB=$(find $OUT/${TAG}_sq -name "*.db" | head -1)
python3 - "$DB" <<'PY' | tee $OUT/${TAG}_sq_counters.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"))
by = {}
for n, c, k, v in rows:
    if "mibn" in n: by.setdefault(n.split("(")[0].split("::")[-1], {})[c] = (k, v)
for n, d in by.items():
    wc = d.get("SQ_WAVE_CYCLES", (0, 1))[1] or 1
    print("%-24s launches %5d  " % (n[:24], d.get("SQ_WAVE_CYCLES", (0, 0))[0]) + "  ".join("%s %.3f" % (c.replace("SQ_", ""), v / wc) for c, (k, v) in sorted(d.items()) if c != "SQ_WAVE_CYCLES") + "  (fractions of SQ_WAVE_CYCLES)")
PY
# PMC calibration on a known byte count: tools/ubench/sweep_real (8 MiB in + 8 MiB out per request; 64-byte runs in)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $OUT/${TAG}_calib_$c -o c -- $ROOT/tools/ubench/sweep_real 1024 8 > $OUT/${TAG}_calib_$c.log 2>&1
  DB=$(find $OUT/${TAG}_calib_$c -name "*.db" | head -1)
  python3 - "$DB" $c <<'PY' | tee -a $OUT/${TAG}_pmc_calibration.log
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, k, v in cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name"):
    if "sweep" in n: print("%s %s launches %d avg %.1f KB = %.3f GB per launch; known: 1024 requests x 8 MiB = %.3f GB each way" % (c, n[:48], k, v, v * 1024 / 1e9, 1024 * 8 * 1048576 / 1e9))
PY
done
find $OUT -name "*.db" -delete
cd $ROOT
# few host threads per rank (8 ranks on a small CPU quota): the default bench (adaptive policy: the device planner takes over when
# the host's planning workers bound the pipeline), and the host planner alone beside it
for args in "--threads 1" "--threads 2" "--threads 4" "--threads 8" "--threads 2 --no-adaptive" "--threads 4 --no-adaptive"; do
  timeout 300 python bench.py --steps 3 --warmup 3 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']
        print('%-28s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned requests per step %.0f' % ('$args', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step']))
" | tee -a $OUT/${TAG}_threads.log
done
