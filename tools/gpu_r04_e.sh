#!/bin/bash
# Round-4 session E: the sweep kernel's readout path (a digit dies in stage j: kout = 4) in isolation, before / after the wave-local
# hand-over into stage 4; hardware queues.
TAG=${1:-r04_e}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for dead in -1 0 1 2 3 4; do for b in sweep_real_base sweep_real; do echo "== $b dead stage $dead"; timeout 100 tools/ubench/$b 2048 8 $dead 2>&1 | tail -1; done; done > $OUT/${TAG}_sweep_dead.log 2>&1
for dead in 2 4; do echo "== prof dead stage $dead"; timeout 100 tools/ubench/sweep_real_prof 2048 8 $dead 2>&1 | tail -2; done >> $OUT/${TAG}_sweep_dead.log 2>&1
cat $OUT/${TAG}_sweep_dead.log
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
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "default" | tee -a $OUT/${TAG}_queues.log
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "GPU_MAX_HW_QUEUES=8" | tee -a $OUT/${TAG}_queues.log
  GPU_MAX_HW_QUEUES=2 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "GPU_MAX_HW_QUEUES=2" | tee -a $OUT/${TAG}_queues.log
done
