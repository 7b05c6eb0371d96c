#!/bin/bash
# A/B of engine options on the C3 bench, interleaved and repeated (box-to-box and run-to-run noise is ~2 %).  usage: gpu_r03_d.sh "<set>;<set>;..." [repeats]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/r03_d
mkdir -p $OUT
IFS=';' read -ra SETS <<< "$1"
for rep in $(seq 1 ${2:-2}); do
for args in "${SETS[@]}"; do
  timeout 300 python bench.py --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu --no-configs ${ADAPT:---no-adaptive} $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-44s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps']) + '   ' + '  '.join('%s %.0f' % (k.replace('ve_','').replace('_kernel',''), v['GBps']) for k, v in d['kernels'].items()))
    elif 'rror' in l: print(l.rstrip()[:300])
" | tee -a $OUT/bench_ab.log
done
done
