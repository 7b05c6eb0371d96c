"""Debug aid (round 6, session BE): one request of the bench's C3 stream through the device planner under gpu_emit = 2 with engine options from the command line
(name=value ...)."""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import netspec, sorobn_amd
spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
q, ev, ec = netspec.c3_requests(100, 4, 262144 * 3, 4, seed=1)
i0 = 681765
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    be.engine.set_option(k, float(v))
sl = slice(i0, i0 + 1)
be.engine.set_option("gpu_emit", 0)
host = be.engine.query_fixed(to_var[q[sl]][:, None], to_var[ev[sl]], ec[sl])
be.engine.set_option("gpu_emit", 2)
try:
    dev = be.engine.query_fixed(to_var[q[sl]][:, None], to_var[ev[sl]], ec[sl])
    print(sys.argv[1:], "ok", np.array_equal(dev, host))
except Exception as e:
    print(sys.argv[1:], "FAIL", str(e)[:300])
