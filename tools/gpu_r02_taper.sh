cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_form_vs_chain or grid10x10 or schedule_and" 2>&1 | tail -2
for args in "" "--opt sweep_taper=0" "" "--opt sweep_taper=0"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-26s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps']) + ''.join('  %s %.0f' % (k[3:8], v['GBps']) for k, v in d['kernels'].items()))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
