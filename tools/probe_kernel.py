"""GPU probe: where does the VE kernel's time go?  (run on the MI355X box)
 - whole C3 batch at several batch sizes
 - the heaviest request alone (per-workgroup streaming rate)
 - N copies of the heaviest request (aggregate streaming rate when every workgroup streams)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

spec = netspec.grid_spec(10, 10, 4, seed=0)
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
eng = be.engine
to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
N = int(os.environ.get("PROBE_N", "16384"))
q, ev, ec = netspec.c3_requests(100, 4, N, 4, seed=1)
Q, E = to_var[q][:, None], to_var[ev]


def run(Qs, Es, Cs, label, reps=2):
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.query_fixed(Qs, Es, Cs)
        dt = time.perf_counter() - t0
        s = eng.stats()
    print(f"{label:34s} B={len(Qs):6d} wall {dt*1e3:8.1f} ms kernel {s['kernel_ms']:8.2f} ms plan {s['plan_ms']:7.1f} ms "
          f"h2d {s['h2d_ms']:6.1f} bytes {s['alg_bytes']/1e9:8.2f} GB -> {s['alg_bytes']/s['kernel_ms']/1e6:8.1f} GB/s "
          f"steps {s['n_steps']:.0f} wgs {s['n_workgroups']:.0f} arena {s['arena_bytes']/1e9:.1f} GB", flush=True)
    if os.environ.get("PROBE_KSTATS"):
        for k in sorted(eng.kernel_stats(), key=lambda k: -k["ms"])[:6]:
            print(f"      {k['name']:28s} launches {k['launches']:5.0f} items {k['items']:9.0f} ms {k['ms']:8.2f} "
                  f"bytes {k['alg_bytes']/1e9:8.2f} GB -> {k['alg_bytes']/max(k['ms'],1e-9)/1e6:8.1f} GB/s")
    return s


for B in (1024, 4096, 16384):
    if B <= N:
        run(Q[:B], E[:B], ec[:B], f"C3 stream first {B}")

cost = np.array([eng.plan_stats(Q[i], E[i])["alg_bytes"] for i in range(min(N, 4096))])
order = np.argsort(-cost)
print("cost MB: max %.1f p99 %.1f p90 %.1f median %.1f mean %.1f" % (
    cost.max() / 1e6, np.percentile(cost, 99) / 1e6, np.percentile(cost, 90) / 1e6, np.median(cost) / 1e6, cost.mean() / 1e6))
h = order[0]
for copies in (1, 8, 64, 256, 512, 2048):
    idx = np.full(copies, h)
    run(Q[idx], E[idx], ec[idx], f"heaviest request x{copies}")
m = order[len(order) // 2]
for copies in (1, 2048, 16384):
    idx = np.full(copies, m)
    run(Q[idx], E[idx], ec[idx], f"median request x{copies}")
light = order[-len(order) // 4:]
idx = np.resize(light, 16384)
run(Q[idx], E[idx], ec[idx], "lightest quartile x16384")
