"""GPU probe: wall-clock timeline of the bench's two-deep pipeline (submit / wait per step) after a drained start."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec, sorobn_amd
bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
be = bn.backend; eng = be.engine
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, float(v))
B = 32768; S = 8
qv, ev, ec = netspec.c3_requests(100, 4, (S + 2) * B, 4, seed=1)
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
def sub(s):
    lo, hi = s * B, (s + 1) * B
    return eng.submit_fixed(to_var[qv[lo:hi]][:, None], to_var[ev[lo:hi]], ec[lo:hi])
pend = None
for s in range(2):  # warm-up through the same pipeline (the arena reaches the size of a full chunk: re-allocating it takes seconds)
    nxt = sub(s)
    if pend is not None:
        eng.wait(pend)
    pend = nxt
eng.wait(pend)
eng.drain(); eng.synchronize()
t0 = time.perf_counter(); log = []
pend = None
for s in range(2, 2 + S):
    nxt = sub(s); log.append(("submit", s, time.perf_counter() - t0, eng.stats()["plan_ms"]))
    if pend is not None:
        eng.wait(pend); log.append(("wait", s - 1, time.perf_counter() - t0, 0))
    pend = nxt
eng.wait(pend); log.append(("wait", 2 + S - 1, time.perf_counter() - t0, 0))
eng.drain(); eng.synchronize(); log.append(("drain", 0, time.perf_counter() - t0, 0))
for what, s, t, pl in log:
    print("%-7s step %2d at %8.1f ms  (plan %.1f)" % (what, s, t * 1e3, pl))
print("total %.1f ms for %d steps = %.1f ms/step; kernel_ms total %.1f" % (log[-1][2] * 1e3, S, log[-1][2] * 1e3 / S, 0))
