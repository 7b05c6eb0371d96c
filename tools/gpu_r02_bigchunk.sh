cd "$GRAFT_REPO_ROOT"
for args in "" "--batch 65536 --steps 4 --warmup 1 --opt chunk=32768 --opt arena_gb=220" "--batch 65536 --steps 4 --warmup 1 --opt chunk=24576 --opt arena_gb=160" "--batch 65536 --steps 4 --warmup 1"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-84s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps']) + ''.join('  %s %.0f' % (k[3:8], v['GBps']) for k, v in d['kernels'].items()))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
