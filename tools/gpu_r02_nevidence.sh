# the C3 stream with 1 / 2 / 4 / 8 / 16 evidence nodes per request
cd "$GRAFT_REPO_ROOT"
for ne in 1 2 4 8 16; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --n-evidence $ne 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('n_evidence %2d: %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f  MB/query %.2f' % ($ne, d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps'], r['alg_bytes_per_query'] / 1e6))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
