#!/bin/bash
# Round-2 GPU box session: GPU parity tests, the bench line, rocprofv3 trace + PMC passes.  usage: tools/gpu_r02_session.sh <tag>
TAG=${1:-r02_a}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)") > $OUT/${TAG}_host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -25 $OUT/${TAG}_pytest_gpu.log
timeout 600 python bench.py > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
tail -c 6000 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
tail -20 $OUT/${TAG}_profile_session.log
