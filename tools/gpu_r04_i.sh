#!/bin/bash
# Round-4 session I: the device planner with a factor capacity of 16 axes (-DMIBN_DEVICE_RAW_AXES=16: 14 KB of scratch per lane
# instead of 29 KB, smaller factor records) against the default build: its parity tests, the two-thread rank, the planner alone.
TAG=${1:-r04_i}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
MIBN_LIB=$ROOT/sorobn_amd/libmibn_v_axes16.so timeout 600 python -m pytest tests -m gpu -q -x -k "device_planner or adaptive or device_order or stratified" > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log; tail -5 $OUT/${TAG}_pytest_gpu.log
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; k = d['kernels'].get('order_kernel+emit_kernel', {})
        print('%-64s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned per step %.0f  planner kernels %.1f ms x%d' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step'], k.get('ms', 0) / max(1, k.get('launches', 1)) * 2, k.get('launches', 0)))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
for lib in libmibn.so libmibn_v_axes16.so; do
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 3 --warmup 3 --no-cpu --no-configs --threads 2 2>&1 | summ "$lib --threads 2" | tee -a $OUT/${TAG}_ab.log
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-configs --no-adaptive --threads 2 --opt gpu_emit=1 --opt emit_share=1 2>&1 | summ "$lib --threads 2, everything on the device" | tee -a $OUT/${TAG}_ab.log
done
done
MIBN_LIB=$ROOT/sorobn_amd/libmibn_v_axes16.so timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-configs --no-adaptive --sync --threads 2 --opt gpu_emit=1 --opt emit_share=1 2>&1 | summ "axes16, blocking calls (planner kernels alone)" | tee -a $OUT/${TAG}_ab.log
MIBN_LIB=$ROOT/sorobn_amd/libmibn.so timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-configs --no-adaptive --sync --threads 2 --opt gpu_emit=1 --opt emit_share=1 2>&1 | summ "default, blocking calls (planner kernels alone)" | tee -a $OUT/${TAG}_ab.log
# the few-threads table again (the share controller no longer learns from the tail chunk of a call; steps of 2^18 requests)
for args in "--threads 1" "--threads 2" "--threads 4" "--threads 8" ""; do
  timeout 300 python bench.py --steps 4 --warmup 3 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']
        print('%-28s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned requests per step %.0f  all kernels %.0f GB/s' % ('$args', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step'], d['roofline']['all_kernels_GBps']))
" | tee -a $OUT/${TAG}_threads.log
done
for args in "--threads 2" "--threads 4"; do
  MIBN_LIB=$ROOT/sorobn_amd/libmibn_v_axes16.so timeout 300 python bench.py --steps 4 --warmup 3 --no-cpu --no-configs $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']
        print('%-28s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned requests per step %.0f  all kernels %.0f GB/s' % ('axes16 $args', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config']['device_planned_requests_per_step'], d['roofline']['all_kernels_GBps']))
" | tee -a $OUT/${TAG}_threads.log
done
