"""Per class of work: one C3 chunk with one launch per (level, class) - time, algorithmic bytes and rate of every class
(the table behind DESIGN 4.1 / 4.5).  PROBE_OPTS="sweep=0,..." sets engine options; PROBE_N = requests (default 16384)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

bn = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
eng = bn.backend.engine
to_var = np.array([bn.backend.flat.id[f"{i:03d}"] for i in range(100)], np.int32)
n = int(os.environ.get("PROBE_N", "16384"))
q, ev, ec = netspec.c3_requests(100, 4, n, 4, seed=1)
eng.set_option("chunk", n)
for kv in os.environ.get("PROBE_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, float(v))
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)  # warm-up (clocks, buffers)
eng.set_option("split_kinds", 1)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)
ks = sorted(eng.kernel_stats(), key=lambda k: -k["ms"])
tot_ms = sum(k["ms"] for k in ks); tot_b = sum(k["alg_bytes"] for k in ks)
print("%-28s %8s %9s %7s %7s %9s" % ("class", "launches", "ms", "% time", "% bytes", "GB/s"))
for k in ks:
    print("%-28s %8d %9.3f %7.1f %7.1f %9.0f" % (k["name"], k["launches"], k["ms"], 100 * k["ms"] / tot_ms, 100 * k["alg_bytes"] / tot_b, k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6))
print("%-28s %8s %9.3f %7s %7s %9.0f   (%.2f MB per query)" % ("all (one class at a time)", "", tot_ms, "", "", tot_b / tot_ms / 1e6, tot_b / n / 1e6))
eng.set_option("split_kinds", 0)
eng.query_fixed(to_var[q][:, None], to_var[ev], ec)
s = eng.stats()
print("one launch per level and kernel: kernel %.3f ms -> %.0f GB/s" % (s["kernel_ms"], s["alg_bytes"] / s["kernel_ms"] / 1e6))
