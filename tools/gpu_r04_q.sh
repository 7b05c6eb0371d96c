#!/bin/bash
# Round-4 session Q: work-item sizes again at the larger launches (calls of 52 429 requests): tiles per sweep workgroup, traffic per tile of the level kernel.
TAG=${1:-r04_q}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-36s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for args in "" "--opt sweep_iters=16" "--opt tile_kb=1024" "--opt tile_kb=256" "--opt sweep_adapt=8192"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | summ "default $args" | tee -a $OUT/${TAG}_items.log
done
done
