"""GPU probe: N copies of the heaviest C3 request (uniform big launches: kernel efficiency without scheduling effects)
and the C3 mix, per class of work (split_kinds=1) and as one launch per level."""
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
q, ev, ec = netspec.c3_requests(100, 4, 16384, 4, seed=1)
Q, E = to_var[q][:, None], to_var[ev]
for kv in os.environ.get("PROBE_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, float(v))


def run(Qs, Es, Cs, label, split):
    eng.set_option("split_kinds", split)
    for _ in range(2):
        t0 = time.perf_counter()
        eng.query_fixed(Qs, Es, Cs)
        dt = time.perf_counter() - t0
        s = eng.stats()
    print(f"{label:30s} split={split} B={len(Qs):6d} wall {dt*1e3:8.1f} kernel {s['kernel_ms']:8.2f} ms launches {s['n_launches']:4.0f} bytes {s['alg_bytes']/1e9:8.2f} GB "
          f"-> {s['alg_bytes']/s['kernel_ms']/1e6:8.1f} GB/s", flush=True)
    if split:
        for k in sorted(eng.kernel_stats(), key=lambda k: -k["ms"])[:int(os.environ.get("PROBE_TOP", "6"))]:
            print(f"      {k['name']:32s} launches {k['launches']:5.0f} wgs {k['items']:9.0f} ms {k['ms']:8.2f} "
                  f"{k['alg_bytes']/1e9:8.2f} GB -> {k['alg_bytes']/max(k['ms'],1e-9)/1e6:8.1f} GB/s")


h = 2517  # heaviest of the first 4096 (tools: 393 MB unfused)
idx = np.full(int(os.environ.get("PROBE_COPIES", "1024")), h)
run(Q[idx], E[idx], ec[idx], "heaviest request copies", 1)
run(Q[idx], E[idx], ec[idx], "heaviest request copies", 0)
run(Q, E, ec, "C3 mix", 1)
run(Q, E, ec, "C3 mix", 0)
