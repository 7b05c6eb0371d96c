#!/bin/bash
# Round-4 closing session on the final code (engine calls sized from the shard: 5 x 52 429 requests per 2^18-request step at N = 1):
# the GPU suite, the bench line as the driver runs it, the rocprofv3 trace + PMC passes of the same command, smoke.
TAG=${1:-r04_o}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)"; rocm-smi --showmeminfo vram 2>/dev/null | grep Total) > $OUT/${TAG}_host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
grep -E "passed|failed|rc" $OUT/${TAG}_pytest_gpu.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
echo "bench rc $?"; tail -c 400 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
tail -9 $OUT/${TAG}_profile_session.log | cut -c1-300
find $OUT -name "*.db" -delete
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
