#!/bin/bash
# device order search: parity test, then the bench with few / default planner threads, host vs device search
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "device_order_search" 2>&1 | tail -5
for args in "--threads 2" "--threads 3" "--threads 4" "--threads 8" ""; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-52s %.0f q/s  ms/step %.1f  plan %.1f kernel %.1f  MB/query %.2f  GB/s %.0f' % ('$args', d['value'], d['ms_per_step'], b['plan_ms'], b['kernel_ms'], r['alg_bytes_per_query']/1e6, r['achieved']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done 2>&1 | tee gpurun_out/r02_e_gpu_search.log
