#!/bin/bash
# Round-4 session T: the adaptive policy's switch-off share (0.3 -> 0.2: with the controller's floor of 0.25 a host-bound rank keeps the
# device planner) at 6 and 8 planning threads - at 6 the policy oscillated (219 k queries/s, below the 247 k of a 4-thread rank).
TAG=${1:-r04_t}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for args in "--threads 6" "--threads 8" "--threads 5"; do
  MIBN_LIB=$ROOT/sorobn_amd/libmibn_v_off02.so timeout 300 python bench.py --steps 4 --warmup 4 --no-cpu --no-configs --batch 32768 $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']
        print('%-22s %-14s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned requests per step %.0f' % ('off-share 0.2', '$args', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step']))
" | tee -a $OUT/${TAG}_policy.log
done
