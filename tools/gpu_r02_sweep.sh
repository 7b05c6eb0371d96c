# SWEEP form on the device: parity first, then the C3 rate with and without it
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or grid10x10 or c3_stream or heavy or chain_form or wide_grids" 2>&1 | tail -15
for args in "" "--opt sweep=0" "--opt sweep=4" ""; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-16s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f  MB/query %.2f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps'], r['alg_bytes_per_query'] / 1e6))
        for k, v in d['kernels'].items(): print('      %-20s launches %5d ms %8.1f  GB %8.1f  -> %6.0f GB/s' % (k, v['launches'], v['ms'], v['alg_GB'], v['GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
