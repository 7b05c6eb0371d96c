# SQ / instruction counters and phase timers of wave_plan_kernel alone (tools/ubench/wave_plan_bench[_prof]): usage tools/gpu_wave_counters.sh <tag>
TAG=${1:-wave}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 120 tools/ubench/wave_plan_bench_prof 32768 4 512 > $OUT/${TAG}_wave_prof.log 2>&1
timeout 120 tools/ubench/wave_plan_bench_prof 32768 16 512 >> $OUT/${TAG}_wave_prof.log 2>&1
( cd /tmp; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES -d $OUT/${TAG}_insts -o c -- $GRAFT_REPO_ROOT/tools/ubench/wave_plan_bench 32768 4 64 > $OUT/${TAG}_insts.log 2>&1 )
python3 - "$(find $OUT/${TAG}_insts -name '*.db' | head -1)" <<'PY' | tee $OUT/${TAG}_inst_counts.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, k, v in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
    if "wave_plan" in n: print("%-20s %-18s rows %4d  total %.4g  per request (4 launches x 32768) %.1f" % (n[:20], c, k, v, v / (4 * 32768.0)))
PY
find $OUT -name "*.db" -delete
( cd /tmp; timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_ic -o c -- $GRAFT_REPO_ROOT/tools/ubench/wave_plan_bench 32768 4 64 >> $OUT/${TAG}_insts.log 2>&1 )
python3 - "$(find $OUT/${TAG}_ic -name '*.db' | head -1)" <<'PY' | tee -a $OUT/${TAG}_inst_counts.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, k, v in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
    if "wave_plan" in n: print("%-20s %-18s rows %4d  total %.4g  per request (4 launches x 32768) %.1f" % (n[:20], c, k, v, v / (4 * 32768.0)))
PY
find $OUT -name "*.db" -delete
