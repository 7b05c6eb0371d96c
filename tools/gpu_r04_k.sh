#!/bin/bash
# Round-4 session K: hardware queues again (GPU_MAX_HW_QUEUES 4 = default against 8), four alternating pairs.
TAG=${1:-r04_k}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-28s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps']))
"; }
for rep in 1 2 3 4; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "default" | tee -a $OUT/${TAG}_queues.log
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "GPU_MAX_HW_QUEUES=8" | tee -a $OUT/${TAG}_queues.log
done
