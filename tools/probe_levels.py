"""One C3 chunk, one launch per level, the engine's launch trace on stderr (per-level times)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
be = bn.backend
eng = be.engine
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
q, ev, ec = netspec.c3_requests(100, 4, 16384, 4, seed=1)
eng.set_option("chunk", 16384)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)  # warm-up (allocations)
eng.set_option("trace", 1)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)
