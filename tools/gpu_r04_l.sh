#!/bin/bash
# Round-4 session L: staggered levels (option stagger: G groups of requests whose programs start G-th of the level count apart, so that
# every level mixes the level-kernel-heavy start of a program with the sweep-heavy middle of another) - re-measured now that the launches of a
# level really overlap.
TAG=${1:-r04_l}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-28s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for args in "" "--opt stagger=2" "--opt stagger=3" "--opt stagger=4"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | summ "default $args" | tee -a $OUT/${TAG}_stagger.log
done
done
