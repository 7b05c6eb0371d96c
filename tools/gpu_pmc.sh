#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq1 -o p -- python $ROOT/tools/probe_pmc.py > $OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d $OUT/pmc_sq2 -o p -- python $ROOT/tools/probe_pmc.py > $OUT/pmc_sq2.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in ("gpurun_out/pmc_sq1", "gpurun_out/pmc_sq2"):
    for db in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        rows = list(cur.execute("select name, counter_name, count(*), sum(counter_value), sum(duration)/1e6 from pmc_events group by name, counter_name order by 2"))
        for n, c, k, v, dms in rows:
            if "ve_level" in n:
                print(f"{c:26s} launches {k:5d} sum {v:16.0f}  kernel-ms {dms:9.2f}")
PY
find $OUT -name "*.db" -size +5M -delete
