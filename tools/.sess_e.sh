cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['pipeline_clocks_ms_per_step']
        print('%-44s %.0f q/s  ms/step %.1f  gpu busy %.1f  planner wall %.1f  device-planned %.0f  all kernels %.0f GB/s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], b['planner_wall_ms_inside_submit_calls'], d['config'].get('device_planned_requests_per_step', 0), r['all_kernels_GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
mkdir -p gpurun_out
for a in "--n-evidence 16 --steps 10 --warmup 4" "--threads 6 --steps 6 --warmup 4" "--threads 8 --steps 6 --warmup 4" "--steps 6 --warmup 3"; do
  timeout 300 python bench.py --no-cpu --no-configs $a 2>&1 | summ "adaptive: $a" | tee -a gpurun_out/r05_e_policy.log
done
AB="--opt tile_kb=512;--opt tile_kb=1024;--opt tile_kb=256;--opt sweep_iters=16;--opt sweep_iters=4" LIBS="libmibn.so" REPS=2 STEPS=5 bash tools/gpu_session.sh r05_e ab
AB="" LIBS="libmibn.so;libmibn_v_lw2.so" REPS=3 STEPS=5 bash tools/gpu_session.sh r05_e2 ab
