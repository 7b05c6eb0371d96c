cd "$GRAFT_REPO_ROOT"
for args in "" "--opt chunk_sets=3" "" "--opt chunk_sets=3" "" "--opt chunk_sets=3" "--opt chunk_sets=3 --opt first_chunk=0"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-40s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
