#!/bin/bash
# Round-4 session A: the GPU parity suite with the new tests (stratified C3, n_evidence variants, sampling walk, dry run of config 4),
# the bench line as the driver runs it, and the A/B of the repaired `overlap` option (it used to serialise).
TAG=${1:-r04_a}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)"; rocm-smi --showtopo 2>/dev/null | head -20) > $OUT/${TAG}_host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -30 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
echo "bench rc $?"; tail -c 3000 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-34s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f GB/s x%d' % (k[:14], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
"; }
for rep in 1 2; do
for args in "--opt overlap=1" "--opt overlap=0"; do
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-configs $args 2>/dev/null | summ "$args" | tee -a $OUT/${TAG}_overlap.log
done
done
PROBE_N=32768 timeout 300 python tools/probe_classes.py > $OUT/${TAG}_probe_classes.log 2>&1; tail -30 $OUT/${TAG}_probe_classes.log
timeout 300 python tools/probe_launches.py 2>&1 | grep "mibn launch" > $OUT/${TAG}_launches.log; wc -l $OUT/${TAG}_launches.log
