#!/bin/bash
# Round-4 session C: the sweep kernel's wave-local tail (two workgroup barriers per tile instead of three) and the vmcnt(8) wait,
# A/B as library variants (MIBN_LIB): libmibn_v_base.so = neither, libmibn_v_tail.so = tail only, libmibn.so = both;
# the repaired gibbs_kernel8; the whole GPU suite.
TAG=${1:-r04_c}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# isolated sweep kernel (tools/ubench/sweep_real.hip: identical five-variable OWN steps) + phase timers
for b in sweep_real_base sweep_real_tail sweep_real sweep_real_base sweep_real_tail sweep_real; do echo "== $b"; timeout 100 tools/ubench/$b 2048 8 2>&1 | tail -3; done > $OUT/${TAG}_sweep_real.log 2>&1
for b in sweep_real_prof_base sweep_real_prof; do echo "== $b"; timeout 100 tools/ubench/$b 2048 8 2>&1 | tail -14; done > $OUT/${TAG}_sweep_prof.log 2>&1
cat $OUT/${TAG}_sweep_real.log $OUT/${TAG}_sweep_prof.log
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-44s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
"; }
for rep in 1 2; do
for lib in libmibn_v_base.so libmibn_v_tail.so libmibn.so; do
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs 2>/dev/null | summ "$lib" | tee -a $OUT/${TAG}_ab.log
  MIBN_LIB=$ROOT/sorobn_amd/$lib timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs --opt overlap=0 2>/dev/null | summ "$lib overlap=0" | tee -a $OUT/${TAG}_ab.log
done
done
for args in "--opt sweep_iters=4" "--opt sweep_iters=16"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>/dev/null | summ "libmibn.so $args" | tee -a $OUT/${TAG}_ab.log
done
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -16 $OUT/${TAG}_pytest_gpu.log
python - <<'PY' 2>&1 | tee $OUT/${TAG}_gibbs.log
import sys, time
sys.path.insert(0, "tests")
import numpy as np, netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(5, 10, 8, seed=0), sorobn_amd.BayesNet).use_device(0)
rng = np.random.default_rng(1)
ev = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
exact = bn.query("025", event=ev).to_numpy()
eng = bn.backend.engine
for mode in (1, 2):
    eng.set_option("gibbs_lds", mode)
    bn.query("025", event=ev, algorithm="gibbs", n_iterations=1000, n_chains=128)
    for chains in (128, 1024, 4096):
        got = bn.query("025", event=ev, algorithm="gibbs", n_iterations=100_000, n_chains=chains).to_numpy()
        print("gibbs_lds=%d chains %5d: kernel %.1f ms, %.3f us per update and chain, max|err| vs exact %.2e" %
              (mode, chains, eng.stats()["kernel_ms"], eng.stats()["kernel_ms"] * 1e3 / 100_000, float(np.max(np.abs(got - exact)))))
PY
