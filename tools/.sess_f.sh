cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['pipeline_clocks_ms_per_step']; k = d['kernels'].get('order_kernel+emit_kernel', {})
        print('%-52s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned %.0f  all kernels %.0f GB/s  planner kernels %.1f ms per pair' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config'].get('device_planned_requests_per_step', 0), r['all_kernels_GBps'], k.get('ms', 0) / max(1, k.get('launches', 2)) * 2))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
mkdir -p gpurun_out
for a in "" "--opt plan_lanes=16" "--opt plan_lanes=24" "--opt plan_lanes=48" "--opt plan_lanes=64" "--opt plan_waves=8" "--opt plan_lanes=16 --opt plan_waves=8"; do
  timeout 300 python bench.py --no-cpu --no-configs --threads 2 --steps 5 --warmup 4 $a 2>&1 | summ "threads 2 $a" | tee -a gpurun_out/r05_f_planlanes.log
done
for t in 4 5; do timeout 300 python bench.py --no-cpu --no-configs --threads $t --steps 5 --warmup 4 2>&1 | summ "threads $t" | tee -a gpurun_out/r05_f_planlanes.log; done
