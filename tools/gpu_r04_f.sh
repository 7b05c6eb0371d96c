#!/bin/bash
# Round-4 session F: the wave-owned tail for five-variable passes in which one digit dies (kout = 4): isolated (sweep_real with a
# dying stage), the GPU suite, A/B against libmibn_v_nodead.so (-DMIBN_SWEEP_DEAD_TAIL=0).
TAG=${1:-r04_f}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for dead in -1 0 1 2 3 4; do echo "== sweep_real dead stage $dead"; timeout 100 tools/ubench/sweep_real 2048 8 $dead 2>&1 | tail -1; done > $OUT/${TAG}_sweep_dead.log 2>&1
cat $OUT/${TAG}_sweep_dead.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -12 $OUT/${TAG}_pytest_gpu.log
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-40s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for lib in libmibn_v_nodead.so libmibn.so; do
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "$lib" | tee -a $OUT/${TAG}_ab.log
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs --opt overlap=0 2>&1 | summ "$lib overlap=0" | tee -a $OUT/${TAG}_ab.log
done
done
