#!/bin/bash
# Round-4 session B: the segment kernel (third stream), overlap inside lanes, the eight-lanes-per-chain Gibbs kernel.
TAG=${1:-r04_b}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_pytest_gpu.log
tail -16 $OUT/${TAG}_pytest_gpu.log
# Gibbs: config 5 (5x10 grid, K = 8), 100 k updates x 128 / 1024 chains: eight lanes per chain (gibbs_lds=1) against one chain per lane (2)
python - <<'PY' 2>&1 | tee $OUT/${TAG}_gibbs.log
import sys, time
sys.path.insert(0, "tests")
import numpy as np, netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(5, 10, 8, seed=0), sorobn_amd.BayesNet).use_device(0)
rng = np.random.default_rng(1)
ev = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
exact = bn.query("025", event=ev).to_numpy()
eng = bn.backend.engine
for mode in (1, 2, 1, 2):
    eng.set_option("gibbs_lds", mode)
    bn.query("025", event=ev, algorithm="gibbs", n_iterations=1000, n_chains=128)
    for chains in (128, 1024, 4096):
        t0 = time.perf_counter()
        got = bn.query("025", event=ev, algorithm="gibbs", n_iterations=100_000, n_chains=chains).to_numpy()
        dt = time.perf_counter() - t0
        print("gibbs_lds=%d chains %5d: wall %.1f ms, kernel %.1f ms, %.3f us per update and chain, max|err| vs exact %.2e" %
              (mode, chains, dt * 1e3, eng.stats()["kernel_ms"], eng.stats()["kernel_ms"] * 1e3 / 100_000, float(np.max(np.abs(got - exact)))))
PY
# lanes + overlap: same answers
python - <<'PY' 2>&1 | tee $OUT/${TAG}_lanes_check.log
import sys
sys.path.insert(0, "tests")
import numpy as np, netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet).use_device(0)
be = bn.backend
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
q, ev, ec = netspec.c3_requests(100, 4, 98304, 4, seed=1)
ref = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec)
for opts in ({"streams": 2}, {"streams": 2, "chunk": 16384}, {"streams": 1, "seg_kernel": 0}, {"streams": 1, "overlap": 0}):
    for k, v in opts.items(): be.engine.set_option(k, v)
    got = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec)
    print(opts, "bit-identical:", bool(np.array_equal(ref, got)))
PY
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['pipeline_clocks_ms_per_step']; r = d['roofline']
        print('%-44s %.0f q/s  ms/step %.1f  gpu busy %.1f  all kernels %.0f GB/s  %s' % ('$1', d['value'], d['ms_per_step'], b['gpu_busy_ms'], r['all_kernels_GBps'],
              '  '.join('%s %.0f x%d' % (k[:18], v['GBps'], v['launches']) for k, v in d['kernels'].items())))
"; }
for rep in 1 2; do
for args in "" "--opt seg_kernel=0" "--opt streams=2" "--opt streams=2 --opt chunk=16384" "--opt overlap=0" "--opt streams=2 --opt overlap=0"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu --no-configs $args 2>/dev/null | summ "default $args" | tee -a $OUT/${TAG}_ab.log
done
done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
echo "bench rc $?"; tail -c 1500 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
