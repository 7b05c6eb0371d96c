cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for args in "--batch 65536 --opt chunk=16384" "--batch 65536 --opt chunk=32768" "--batch 65536 --opt arena_gb=240 --opt chunk=65536" "--batch 32768 --opt chunk=32768" "--batch 32768"; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-52s %.0f q/s  ms/step %.1f  plan %.1f kernel %.1f  GB/s %.0f launches %d ms/launch %.2f' % ('$args', d['value'], d['ms_per_step'], b['plan_ms'], b['kernel_ms'], r['achieved'], r['launches'], r['ms_per_launch']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
done
