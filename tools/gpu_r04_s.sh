#!/bin/bash
# Round-4 session S: the adaptive policy's host-bound threshold (planner wall time over GPU kernel time: 1.15 against 1.03) for ranks with
# 6 / 8 / 12 planning threads - at 8 threads round 4's table shows 240 k queries/s with the planning 14 % above the kernel time and the device idle.
TAG=${1:-r04_s}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for lib in libmibn.so libmibn_v_hb103.so; do
for args in "--threads 8" "--threads 6" "--threads 12"; do
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 4 --warmup 4 --no-cpu --no-configs --batch 32768 $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']
        print('%-22s %-14s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned requests per step %.0f' % ('$lib', '$args', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step']))
" | tee -a $OUT/${TAG}_policy.log
done
done
