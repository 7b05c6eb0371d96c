#!/bin/bash
# staggered-level experiment: bench at several group counts
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for s in 1 2 3 4 6; do
  python bench.py --steps 5 --warmup 2 --no-cpu --no-configs --opt stagger=$s 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print('stagger $s: %.0f q/s  ms/step %.1f  kernel_ms/step %.1f  launches %d  ms/launch %.3f  GB/s %.0f frac %.3f' % (d['value'], d['ms_per_step'], d['breakdown_ms_per_step']['kernel_ms'], r['launches'], r['ms_per_launch'], r['achieved'], r['frac']))
"
done 2>&1 | tee gpurun_out/r02_b_stagger.log
