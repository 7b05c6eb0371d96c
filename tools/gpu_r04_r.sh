#!/bin/bash
# Round-4 session R: the sweep kernel without its spill (the readout paths re-derive the lane's address part per tile): parity + bench.
TAG=${1:-r04_r}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "sweep or golden or stratified or wide_grids or n_evidence or first16 or heavy or smoke" > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log; grep -E "passed|failed|rc" $OUT/${TAG}_pytest_gpu.log | tail -2
for rep in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % (d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'], '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
" | tee -a $OUT/${TAG}_bench.log
done
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs --opt overlap=0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print('overlap=0: %.0f q/s  all kernels %.0f GB/s  %s' % (d['value'], r['all_kernels_GBps'], '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
" | tee -a $OUT/${TAG}_bench.log
