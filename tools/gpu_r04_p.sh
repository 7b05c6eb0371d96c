#!/bin/bash
# Round-4 session P: what the fixed cost per level is - the idle gaps between consecutive levels in rocprofv3's kernel trace of the bench
# command (tools/rocprof_summary.py: level groups = connected components of the VE kernels' intervals).
TAG=${1:-r04_p}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for args in "" "--batch 32768"; do
  rm -rf $OUT/${TAG}_trace
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-adaptive $args > $OUT/${TAG}_trace.log 2>&1
  T=$(find $OUT/${TAG}_trace -name "*.db" | head -1)
  echo "== bench.py $args" >> $OUT/${TAG}_gaps.txt
  python $ROOT/tools/rocprof_summary.py $T 2>&1 | grep -i "idle gaps\|^levels\|LevelArgs)  " | cut -c1-400 >> $OUT/${TAG}_gaps.txt
done
find $OUT -name "*.db" -delete
cat $OUT/${TAG}_gaps.txt
