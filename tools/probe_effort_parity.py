"""GPU probe (round 6): the wave planner against the host planner under gpu_emit = 2 (every program word, work item and statistic compared by the
engine) with order_effort 1 and the second emission on the device, on long C3 streams of 1 / 4 / 8 / 16 evidence nodes; then second_above = 0
(every request emits two programs).   python tools/probe_effort_parity.py [requests per stream]"""
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import netspec
import sorobn_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
be.engine.set_option("second_on_device", 1)
for above, n_evs in ((2e7, (4, 1, 8, 16)), (0.0, (4, 16))):
    be.engine.set_option("second_above", above)
    for n_ev in n_evs:
        q, ev, ec = netspec.c3_requests(100, 4, n, n_ev, seed=40 + n_ev)
        be.engine.set_option("gpu_emit", 0)
        host = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec)
        hb = be.engine.stats()["alg_bytes"]
        be.engine.set_option("gpu_emit", 2)
        t0 = time.time()
        dev = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec)
        planned = [k for k in be.engine.kernel_stats() if k["name"] == "order_kernel+emit_kernel"]
        print(f"second_above {above:g}, n_evidence {n_ev}: {n} requests, {int(planned[0]['items']) if planned else 0} planned by the device and compared word for word "
              f"({time.time() - t0:.1f} s), posteriors bit for bit: {np.array_equal(dev, host)}, {hb / n / 1e6:.2f} MB per query", flush=True)
