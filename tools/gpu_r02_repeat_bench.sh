cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-configs 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%.0f q/s  ms/step %.1f  plan %.1f kernel %.1f  GB/s %.0f launches %d' % (d['value'], d['ms_per_step'], b['plan_ms'], b['kernel_ms'], r['achieved'], r['launches']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
