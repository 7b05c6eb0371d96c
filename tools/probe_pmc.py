"""Workload for PMC passes: N copies of the heaviest C3 request, one launch per level (80 % cx16/nc16 FIBER tiles)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
eng = bn.backend.engine
to_var = np.array([bn.backend.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
q, ev, ec = netspec.c3_requests(100, 4, 4096, 4, seed=1)
idx = np.full(int(os.environ.get("PROBE_COPIES", "1024")), 2517)
for _ in range(2):
    eng.query_fixed(to_var[q][idx][:, None], to_var[ev][idx], ec[idx])
s = eng.stats()
print(f"kernel {s['kernel_ms']:.2f} ms {s['alg_bytes']/s['kernel_ms']/1e6:.1f} GB/s")
