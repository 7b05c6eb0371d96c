"""GPU probe: where do ve_sweep_dma_kernel and ve_sweep_kernel disagree?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec, sorobn_amd
spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
q, ev, ec = netspec.c3_requests(100, 4, 3072, 4, seed=5)
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
for opts in ({"sweep": 5}, {"sweep": 4}, {"sweep": 3}, {"sweep": 5, "sweep_canon": 0}):
    for k, v in {"sweep": 5, "sweep_canon": 1, **opts}.items():
        be.engine.set_option(k, v)
    res = []
    for dma in (0, 1, 0, 1):
        be.engine.set_option("sweep_dma", dma)
        res.append(be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec))
    d01 = np.abs(res[0] - res[1]); d02 = np.abs(res[0] - res[2]); d13 = np.abs(res[1] - res[3])
    print(opts, "old vs new: max %.2e, requests differing %d of %d; old vs old %.1e; new vs new %.1e" % (d01.max(), int((d01.max(1) > 0).sum()), len(q), d02.max(), d13.max()), flush=True)
    bad = np.nonzero(d01.max(1) > 0)[0][:5]
    for b in bad:
        print("   request", b, "q", q[b], "ev", ev[b], "diff", d01[b])
