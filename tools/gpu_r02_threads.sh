cd "$GRAFT_REPO_ROOT"
cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -i thrott
for args in "" "--threads 16" "--threads 24" "--threads 32" "--threads 16 --opt chunk_sets=3" "--threads 32 --opt chunk_sets=3" "--threads 128"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-40s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -i thrott
