#!/bin/bash
# Round-4 session M: larger engine calls / chunks with steps of 2^18 requests (whole calls: 6 x 43 691, 5 x 52 429) against 8 x 32 768;
# sweep_min 3 (pair steps back on the MFMA class) now that the launches of a level overlap.
TAG=${1:-r04_m}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-64s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  MB/query %.2f  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'], r['alg_bytes_per_query'] / 1e6,
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for args in "" "--batch 43691 --opt chunk=43691 --opt arena_gb=230" "--batch 52429 --opt chunk=52429 --opt arena_gb=250" "--opt sweep_min=3"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | summ "default $args" | tee -a $OUT/${TAG}_chunk.log
done
done
