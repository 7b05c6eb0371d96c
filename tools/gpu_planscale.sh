#!/bin/bash
# planner scaling on the GPU box's host (CPU only)
make -C oracle -s
for t in 1 8 32 64 128; do echo "--- 1 process x $t threads"; python tools/planner_load.py $t 6; done
for np in 2 4 8; do
  echo "--- $np processes x 32 threads"
  pids=()
  for i in $(seq 1 $np); do python tools/planner_load.py 32 8 > gpurun_out/ps_$i.log 2>&1 & pids+=($!); done
  for p in "${pids[@]}"; do wait $p; done
  cat gpurun_out/ps_*.log; rm -f gpurun_out/ps_*.log
done
echo "--- 8 processes x 32 threads, pinned to disjoint cpu ranges"
pids=()
for i in 0 1 2 3 4 5 6 7; do taskset -c $((i*16))-$((i*16+15)),$((128+i*16))-$((128+i*16+15)) python tools/planner_load.py 32 8 > gpurun_out/ps_$i.log 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
cat gpurun_out/ps_*.log; rm -f gpurun_out/ps_*.log
lscpu | grep -E "NUMA|Thread|Core|Socket" 
