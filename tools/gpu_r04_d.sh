#!/bin/bash
# Round-4 session D: (1) tools/ubench/pattern_pad - the sweep tile's access pattern with a padded stride between its 64-byte runs;
# (2) larger chunks (49 152 requests, 240 GB arena budget).
TAG=${1:-r04_d}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 tools/ubench/pattern_pad 2048 > $OUT/${TAG}_pattern_pad.log 2>&1; cat $OUT/${TAG}_pattern_pad.log
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-52s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for args in "" "--batch 49152 --opt chunk=49152 --opt arena_gb=245" "--batch 40960 --opt chunk=40960 --opt arena_gb=220"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>&1 | summ "default $args" | tee -a $OUT/${TAG}_chunk.log
done
done
