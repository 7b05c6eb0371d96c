import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, netspec, sorobn_amd, golden_util as gu
spec = next(n for n in gu.load("examples.json") if n["spec"]["name"] == "asia")["spec"]
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
names = list(be.flat.names)
rng = np.random.default_rng(0)
B = 100000
q = rng.integers(0, 8, B).astype(np.int32)
ne = 2
ev = np.array([rng.permutation([v for v in range(8) if v != q[i]])[:ne] for i in range(B)], np.int32)
ec = rng.integers(0, 2, (B, ne)).astype(np.int32)
for rep in range(3):
    t0 = time.perf_counter(); post = be.engine.query_fixed(q[:, None], ev, ec); dt = time.perf_counter() - t0
    s = be.engine.stats()
    print(f"wall {dt*1e3:.2f} ms  total {s['total_ms']:.2f} plan {s['plan_ms']:.2f} h2d {s['h2d_ms']:.2f} kernel {s['kernel_ms']:.2f} d2h {s['d2h_ms']:.2f} launches {s['n_launches']:.0f}")
