#!/bin/bash
# Round-3 quick A/B: sweep kernel in isolation (phase timers) and in the C3 mix.  usage: tools/gpu_r03_c.sh "<bench arg sets separated by ;>"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/r03_c
mkdir -p $OUT
timeout 100 tools/ubench/sweep_real_prof 2048 8 2>&1 | tee $OUT/sweep_prof.log
timeout 100 tools/ubench/sweep_real 2048 8 2>&1 | tee $OUT/sweep_real.log
IFS=';' read -ra SETS <<< "${1:---opt sweep_dma=0;--opt sweep_dma=1}"
for args in "${SETS[@]}"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-44s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f  MB/query %.2f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps'], r['alg_bytes_per_query'] / 1e6))
        for k, v in d['kernels'].items(): print('      %-20s launches %5d ms %8.1f  GB %8.1f  -> %6.0f GB/s' % (k, v['launches'], v['ms'], v['alg_GB'], v['GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
" | tee -a $OUT/bench_ab.log
done
