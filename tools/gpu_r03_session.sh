#!/bin/bash
# Round-3 GPU box session: GPU parity suite, the bench line as the driver runs it, rocprofv3 trace + PMC passes, PMC calibration of
# the sweep kernels' access pattern, the few-host-threads scenario, the whole 1 M-request stream.  usage: tools/gpu_r03_session.sh <tag>
TAG=${1:-r03_a}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)") > $OUT/${TAG}_host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -25 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --full-stream > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
tail -30 $OUT/${TAG}_profile_session.log
# PMC calibration on a known byte count: tools/ubench/sweep_real (8 MiB in + 8 MiB out per request; 64-byte runs in)
cd /tmp
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
find $OUT -name "*.db" -path "*calib*" -delete
cd $ROOT
# few host threads per rank (8 ranks on a small CPU quota): the default bench (adaptive policy: the device planner - order_kernel +
# emit_kernel - takes over when the host's planning workers bound the pipeline), and the host planner alone beside it
for args in "--threads 1" "--threads 2" "--threads 4" "--threads 8" "--threads 2 --no-adaptive" "--threads 4 --no-adaptive"; do
  timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['breakdown_ms_per_step']
        print('%-28s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  device-planned requests per step %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], d['config']['device_planned_requests_per_step']))
" | tee -a $OUT/${TAG}_threads.log
done
# the device planner at scale: 262 144 requests of the C3 stream planned by the host's workers and by order_kernel + emit_kernel -
# the posteriors must agree bit for bit
python - <<'PY' 2>&1 | tee -a $OUT/${TAG}_threads.log
import sys, time
sys.path.insert(0, "tests")
import numpy as np, netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet).use_device(0)
be = bn.backend
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
q, ev, ec = netspec.c3_requests(100, 4, 262144, 4, seed=1)
t0 = time.perf_counter(); host = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec); t1 = time.perf_counter()
be.engine.set_option("gpu_emit", 1); be.engine.set_option("emit_share", 1.0)
dev = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec); t2 = time.perf_counter()
print("device planner at scale: 262144 requests, posteriors identical bit for bit: %s (host-planned %.2f s, device-planned %.2f s, blocking calls)" % (bool(np.array_equal(host, dev)), t1 - t0, t2 - t1))
PY
