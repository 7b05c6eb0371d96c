import os, sys
import numpy as np
ROOT = os.getcwd()
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
eng = bn.backend.engine
to_var = np.array([bn.backend.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
n = 32768
q, ev, ec = netspec.c3_requests(100, 4, n, 4, seed=1)
eng.set_option("chunk", n); eng.set_option("first_chunk", 0)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)
eng.set_option("trace", 1)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)
