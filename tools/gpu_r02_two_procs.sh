# How much would request-level concurrency give?  Two independent processes (two HIP contexts) on the same GPU, half the planner threads each.
cd "$GRAFT_REPO_ROOT"
run() { timeout 300 python bench.py --steps $2 --warmup 2 --no-cpu --no-configs --no-adaptive --threads 16 $3 2>&1 | python -c "
import sys, json, time
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('$1 %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f   (line printed at %.1f)' % (d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], time.time() % 1000))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
echo "one process:"; run solo 30
echo "two processes at once, 45 GB of arena each:"; run A 30 "--opt arena_gb=45" & run B 30 "--opt arena_gb=45" & wait
echo "one process with 45 GB of arena:"; run solo45 30 "--opt arena_gb=45"
