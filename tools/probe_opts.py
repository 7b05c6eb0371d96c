"""GPU probe: C3 batch under different engine options (tile_h, big_iters, chunk).  PROBE_SETS="tile_h=32;tile_h=64,big_iters=8192" """
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
eng = be.engine
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
N = int(os.environ.get("PROBE_N", "16384"))
q, ev, ec = netspec.c3_requests(100, 4, N, 4, seed=1)
Q, E = to_var[q][:, None], to_var[ev]
for opts in os.environ.get("PROBE_SETS", "tile_h=128").split(";"):
    for kv in opts.split(","):
        k, v = kv.split("=")
        eng.set_option(k, float(v))
    for _ in range(2):
        t0 = time.perf_counter()
        eng.query_fixed(Q, E, ec)
        dt = time.perf_counter() - t0
        s = eng.stats()
    print(f"[{opts}] wall {dt*1e3:8.1f} ms kernel {s['kernel_ms']:8.2f} plan {s['plan_ms']:7.1f} h2d {s['h2d_ms']:6.1f} launches {s['n_launches']:.0f} "
          f"bytes {s['alg_bytes']/1e9:8.2f} GB -> {s['alg_bytes']/s['kernel_ms']/1e6:8.1f} GB/s wgs {s['n_workgroups']:.0f}", flush=True)
    for k in sorted(eng.kernel_stats(), key=lambda k: -k["ms"])[:14]:
        print(f"      {k['name']:28s} launches {k['launches']:5.0f} items {k['items']:9.0f} ms {k['ms']:8.2f} "
              f"bytes {k['alg_bytes']/1e9:8.2f} GB -> {k['alg_bytes']/max(k['ms'],1e-9)/1e6:8.1f} GB/s")
