#!/bin/bash
# Round-3 session f: the whole GPU suite, the default bench line (driver shape), the full-stream figure
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/r03_f
mkdir -p $OUT
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)") > $OUT/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -22 $OUT/pytest_gpu.log
timeout 900 python bench.py --full-stream > $OUT/bench.log 2> $OUT/bench.err; tail -c 3000 $OUT/bench.log; tail -5 $OUT/bench.err
