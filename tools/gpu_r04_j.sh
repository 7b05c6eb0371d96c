#!/bin/bash
# Round-4 closing session on the final code (steps of 2^18 requests): the bench line as the driver runs it and the rocprofv3 trace +
# PMC passes of the same command.
TAG=${1:-r04_j}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)") > $OUT/${TAG}_host.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
echo "bench rc $?"; tail -c 600 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
tail -12 $OUT/${TAG}_profile_session.log
find $OUT -name "*.db" -delete
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
