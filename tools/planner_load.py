"""Host-contention emulator: plans C3 requests on `threads` threads in a loop for `seconds` (CPU only), like one more
bench rank would.  Used to check on a 1-GPU box that 8 ranks' planners fit the host."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import simengine  # noqa: E402
import sorobn_amd  # noqa: E402
from sorobn_amd.flatten import flatten  # noqa: E402

threads, seconds = int(sys.argv[1]), float(sys.argv[2])
period = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0  # seconds per 2 x 16384 requests (0 = flat out)
spec = netspec.grid_spec(10, 10, 4, seed=0)
f = flatten(netspec.build(spec, sorobn_amd.BayesNet))
B = 16384
q, ev, ec = netspec.c3_requests(100, 4, B, 4, seed=2)
to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
Q = np.ascontiguousarray(to_var[q]); E = np.ascontiguousarray(to_var[ev]); EC = np.ascontiguousarray(ec)
L = simengine.lib()
L.plan_sim_bench.restype = C.c_double
p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
hints = np.ascontiguousarray(np.stack(f.hints).reshape(-1), np.int32)
stats = np.zeros(6)
t0 = time.time(); n = 0
while time.time() - t0 < seconds:
    t1 = time.time()
    L.plan_sim_bench(C.c_int32(len(f.card)), p(f.card, C.c_int32), p(f.scope_off, C.c_int64), p(f.scope_vars, C.c_int32),
                     p(f.value_off, C.c_int64), p(f.values, C.c_double), C.c_int32(len(f.hints)), p(hints, C.c_int32),
                     C.c_int64(B), C.c_int32(1), p(Q, C.c_int32), C.c_int32(4), p(E, C.c_int32), p(EC, C.c_int32),
                     C.c_int(threads), p(stats, C.c_double))
    n += 2 * B
    if period:
        time.sleep(max(0.0, period - (time.time() - t1)))
print(f"planner load: {n / (time.time() - t0):.0f} requests/s on {threads} threads")
