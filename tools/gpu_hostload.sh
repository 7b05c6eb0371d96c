#!/bin/bash
# Emulate the host side of an 8-GPU run on a 1-GPU box: 7 planner-only processes with 32 threads each compete with
# a bench rank that is limited to 32 planner threads.
mkdir -p gpurun_out
make -C oracle -s
echo "=== alone, 32 threads"; timeout 300 python bench.py --threads 32 --no-cpu --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms_per_step'])"
pids=()
for i in 1 2 3 4 5 6 7; do python tools/planner_load.py 32 75 ${LOAD_PERIOD:-0.24} > gpurun_out/load_$i.log 2>&1 & pids+=($!); done
sleep 12
echo "=== with 7 x 32-thread planner loads"; timeout 300 python bench.py --threads 32 --no-cpu --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms_per_step'])"
for p in "${pids[@]}"; do wait $p; done
cat gpurun_out/load_*.log
