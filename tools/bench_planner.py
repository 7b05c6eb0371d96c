"""Host planner throughput (CPU only): requests/s for the C3 stream at several thread counts."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import simengine  # noqa: E402
import sorobn_amd  # noqa: E402
from sorobn_amd.flatten import flatten  # noqa: E402

spec = netspec.grid_spec(10, 10, 4, seed=0)
f = flatten(netspec.build(spec, sorobn_amd.BayesNet))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q, ev, ec = netspec.c3_requests(100, 4, B, 4, seed=1)
to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
Q = np.ascontiguousarray(to_var[q]); E = np.ascontiguousarray(to_var[ev]); EC = np.ascontiguousarray(ec)
L = simengine.lib()
L.plan_sim_bench.restype = C.c_double
L.plan_sim_set_chain(int(os.environ.get("CHAIN", "1")))
p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
hints = np.ascontiguousarray(np.stack(f.hints).reshape(-1), np.int32)
for threads in [int(t) for t in (sys.argv[2:] or ["1", "8"])]:
    stats = np.zeros(6)
    best = 1e30
    for _ in range(3):
        ms = L.plan_sim_bench(C.c_int32(len(f.card)), p(f.card, C.c_int32), p(f.scope_off, C.c_int64), p(f.scope_vars, C.c_int32),
                              p(f.value_off, C.c_int64), p(f.values, C.c_double), C.c_int32(len(f.hints)), p(hints, C.c_int32),
                              C.c_int64(B), C.c_int32(1), p(Q, C.c_int32), C.c_int32(4), p(E, C.c_int32), p(EC, C.c_int32),
                              C.c_int(threads), p(stats, C.c_double))
        best = min(best, ms)
    print(f"threads {threads:3d}: {best:8.1f} ms for {B} requests = {best*1e3/B:6.1f} us/request/thread-batch, {B/best*1e3:9.0f} req/s; "
          f"steps {stats[1]:.0f} words/request {stats[2]/B:.0f} bytes/request {stats[0]/B/1e6:.1f} MB; "
          f"build_schedule (1 thread) {stats[3]:.1f} ms, {stats[4]:.0f} items, {stats[5]:.0f} workgroups")
