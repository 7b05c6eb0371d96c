#!/bin/bash
# Round-4 session N: the bench with calls sized from the shard (N = 1: 5 x 52 429 requests per step) and the tapered tail of the sweep
# launches (option sweep_taper: the last T tiles of a level's sweep launch two per workgroup).
TAG=${1:-r04_n}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-44s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  call %d  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'], d['config']['requests_per_engine_call'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for args in "--batch 32768" "" "--opt sweep_taper=1024" "--opt sweep_taper=2048" "--opt sweep_taper=4096"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | summ "default $args" | tee -a $OUT/${TAG}_taper.log
done
done
