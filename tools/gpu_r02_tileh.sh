cd "$GRAFT_REPO_ROOT"
for args in "" "--opt tile_kb=128" "--opt tile_kb=256" "--opt tile_kb=384" "--opt tile_kb=768" "--opt tile_kb=1024" "--opt tile_kb=2048" ""; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-24s %.0f q/s  ms/step %.1f  kernel %.1f  GB/s %.0f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], r['achieved']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
