# PMC calibration of ve_sweep_kernel's access pattern on a known byte count (tools/ubench/sweep_real: 8 MiB in + 8 MiB out per request)
cd "$GRAFT_REPO_ROOT"; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $ROOT/gpurun_out/calib_$c -o c -- $ROOT/tools/ubench/sweep_real 1024 8 > $ROOT/gpurun_out/calib_$c.log 2>&1
  DB=$(find $ROOT/gpurun_out/calib_$c -name "*.db" | head -1)
  python3 - "$DB" $c <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, k, v in cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name"):
    if "sweep" in n: print("%s %s launches %d avg %.1f KB = %.3f GB per launch; known: 1024 requests x 8 MiB = %.3f GB each way" % (c, n[:40], k, v, v * 1024 / 1e9, 1024 * 8 * 1048576 / 1e9))
PY
done
find $ROOT/gpurun_out -name "*.db" -path "*calib*" -delete
