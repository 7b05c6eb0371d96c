# Does the level kernel run faster when the arenas of the requests in flight fit the 256 MiB Infinity Cache?
# (arena_gb = scratch budget of a wave: a chunk is cut into waves of consecutive requests whose arenas fit it)
cd "$GRAFT_REPO_ROOT"
for args in "" "--opt arena_gb=1" "--opt arena_gb=0.5" "--opt arena_gb=0.25" "--opt arena_gb=0.19" "--opt arena_gb=0.125" "--opt arena_gb=0.19 --opt tile_kb=128" "--opt arena_gb=0.06 --opt tile_kb=128"; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-44s %.0f q/s  ms/step %.1f  kernel %.1f  GB/s %.0f  launches %d us/launch %.1f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], r['achieved'], r['launches'], 1e3 * r['ms_per_launch']))
    elif 'rror' in l: print(l.rstrip()[:300])
"
done
