#!/bin/bash
# GPU box session: rocprofv3 kernel trace + PMC passes (separate runs) of the bench command -> gpurun_out/prof_*
# usage: tools/gpu_profile.sh <tag>   (summary written to gpurun_out/<tag>_rocprofv3_summary.txt)
TAG=${1:-prof}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-adaptive"
nproc > $OUT/${TAG}_nproc.txt; lscpu | head -20 >> $OUT/${TAG}_nproc.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace -- $CMD > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o write -- $CMD > $OUT/${TAG}_write.log 2>&1
cd $ROOT
find $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write -name "*.db" | head
T=$(find $OUT/${TAG}_trace -name "*.db" | head -1); F=$(find $OUT/${TAG}_fetch -name "*.db" | head -1); W=$(find $OUT/${TAG}_write -name "*.db" | head -1)
python tools/rocprof_summary.py $T $F $W > $OUT/${TAG}_rocprofv3_summary.txt 2>&1
grep -h '"metric"' $OUT/${TAG}_trace.log | head -1 >> $OUT/${TAG}_rocprofv3_summary.txt
cat $OUT/${TAG}_rocprofv3_summary.txt
# keep the merge-back small: the raw databases are large
find $OUT -name "*.db" -size +20M -delete
