#!/bin/bash
# host-starved regime (the 8-ranks-on-16-CPUs case seen from one rank): 2 planner threads, with and without the adaptive
# planning effort; then the same at the default thread count
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "schedule_and_effort" 2>&1 | tail -2
for args in "--threads 2 --no-adaptive" "--threads 2" "--threads 4 --no-adaptive" "--threads 4" "--no-adaptive" ""; do
  python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-28s %.0f q/s  ms/step %.1f  plan %.1f kernel %.1f  MB/query %.2f  GB/s %.0f' % ('$args', d['value'], d['ms_per_step'], b['plan_ms'], b['kernel_ms'], r['alg_bytes_per_query']/1e6, r['achieved']))
"
done 2>&1 | tee gpurun_out/r02_c_hostbound.log
