#!/bin/bash
# Round-4 session G: stages behind a dying digit skip the fibers that carry nothing (sweep_real with / without, the sweep parity
# tests, A/B against libmibn_v_nodead.so); the segment kernel compiled for 6 / 8 waves per SIMD (more chains in flight, spills).
TAG=${1:-r04_g}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for dead in -1 0 1 2 3 4; do for b in sweep_real_noskip sweep_real; do echo "== $b dead stage $dead"; timeout 100 tools/ubench/$b 2048 8 $dead 2>&1 | tail -1; done; done > $OUT/${TAG}_sweep_dead.log 2>&1
cat $OUT/${TAG}_sweep_dead.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 -k "sweep or golden or stratified or n_evidence or wide_grids or smoke or heavy" > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -8 $OUT/${TAG}_pytest_gpu.log
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
for lib in libmibn_v_nodead.so libmibn.so libmibn_v_seg6.so libmibn_v_seg8.so; do
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | summ "$lib" | tee -a $OUT/${TAG}_ab.log
done
MIBN_LIB=$ROOT/sorobn_amd/libmibn.so timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs --opt overlap=0 2>&1 | summ "libmibn.so overlap=0" | tee -a $OUT/${TAG}_ab.log
done
