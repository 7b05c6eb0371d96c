"""Supplementary measurements of the other BASELINE.json configurations (GPU box): C2 = Asia, 100k exact queries in one
batch; C5 = 50-node K=8 grid, algorithm='gibbs', 100k single-site updates x 128 chains (one GPU's share of 1024).
The bench line proper (bench.py) is C3."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import golden_util as gu  # noqa: E402
import netspec  # noqa: E402
import sorobn_amd  # noqa: E402

out = {}

# ---- C2: Asia, requests per SURVEY 8(d): query var uniform over the 8 nodes, 1-3 evidence nodes, values uniform
spec = next(n for n in gu.load("examples.json") if n["spec"]["name"] == "asia")["spec"]
bn = netspec.build(spec, sorobn_amd.BayesNet)
be = bn.backend
eng = be.engine
n = len(be.flat.card)
rng = np.random.default_rng(0)
B = 100_000
q_off = np.arange(B + 1, dtype=np.int64)
q_vars = rng.integers(0, n, B).astype(np.int32)
ne = rng.integers(1, 4, B)
e_off = np.concatenate([[0], np.cumsum(ne)]).astype(np.int64)
e_vars = np.empty(e_off[-1], np.int32)
for b in range(B):
    others = np.delete(np.arange(n), q_vars[b])
    e_vars[e_off[b]:e_off[b + 1]] = rng.choice(others, ne[b], replace=False)
e_codes = rng.integers(0, 2, e_off[-1]).astype(np.int32)
for _ in range(2):
    t0 = time.perf_counter()
    post, off = eng.query_batch(q_off, q_vars, e_off, e_vars, e_codes)
    dt = time.perf_counter() - t0
s = eng.stats()
sums = np.add.reduceat(post, off[:-1])
out["C2_asia_100k"] = {"queries_per_s": B / dt, "wall_ms": dt * 1e3, "kernel_ms": s["kernel_ms"], "plan_ms": s["plan_ms"],
                       "launches": s["n_launches"], "zero_probability_evidence": int((sums == 0).sum()),
                       "note": "36 CPT numbers: LDS/L2 resident, launch- and host-bound, HBM roofline n/a"}
# single-query latency through the reference-shaped API (config C1 shape)
alarm = next(n for n in gu.load("examples.json") if n["spec"]["name"] == "alarm")["spec"]
bna = netspec.build(alarm, sorobn_amd.BayesNet)
bna.query("Burglary", event={"Mary calls": True, "John calls": True})
t0 = time.perf_counter()
for _ in range(200):
    ans = bna.query("Burglary", event={"Mary calls": True, "John calls": True})
out["C1_alarm_single_query"] = {"ms_per_query": (time.perf_counter() - t0) / 200 * 1e3, "answer": ans.to_numpy().tolist()}

# ---- joint / likelihoods
import pandas as pd  # noqa: E402
t0 = time.perf_counter()
for _ in range(20):
    fjd = bn.full_joint_dist()
out["8f_asia_full_joint_dist"] = {"ms": (time.perf_counter() - t0) / 20 * 1e3, "rows": int(len(fjd)), "sum": float(fjd.sum())}
Xa = fjd.index.to_frame(index=False).iloc[np.random.default_rng(0).integers(0, len(fjd), 100_000)].reset_index(drop=True)
t0 = time.perf_counter()
pp = bn.predict_proba(Xa)
out["8f_asia_predict_proba_100k_rows"] = {"ms": (time.perf_counter() - t0) * 1e3, "mean_log_likelihood": float(np.log(pp.to_numpy()).mean())}
grid = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
Xg = pd.DataFrame(np.random.default_rng(1).integers(0, 4, (100_000, 3)), columns=["011", "055", "090"])
grid.predict_proba(Xg.iloc[:4])
t0 = time.perf_counter()
pg = grid.predict_proba(Xg)
out["8f_grid10x10_predict_proba_3_columns_100k_rows"] = {
    "ms": (time.perf_counter() - t0) * 1e3, "kernel_ms": grid.backend.engine.stats()["kernel_ms"],
    "alg_MB": grid.backend.engine.stats()["alg_bytes"] / 1e6,
    "note": "the reference would need the 4^100-row full joint; here the 97 unobserved variables are eliminated on the device"}

# ---- forward sampling, rejection sampling, likelihood weighting (Asia)
for nsamp in (1_000_000, 16_000_000):
    t0 = time.perf_counter()
    codes = eng.sample(nsamp, seed=3)
    out[f"8f_asia_sample_{nsamp}"] = {"ms": (time.perf_counter() - t0) * 1e3, "samples_per_s": nsamp / (time.perf_counter() - t0)}
del codes
for alg in ("rejection", "likelihood"):
    bn.query("Lung cancer", event={"Smoker": True, "Dispnea": True}, algorithm=alg, n_iterations=1000)
    t0 = time.perf_counter()
    a = bn.query("Lung cancer", event={"Smoker": True, "Dispnea": True}, algorithm=alg, n_iterations=16_000_000)
    out[f"8f_asia_{alg}_16M_samples"] = {"ms": (time.perf_counter() - t0) * 1e3, "samples_per_s": 16e6 / (time.perf_counter() - t0),
                                       "answer": a.to_numpy().tolist()}
out["8f_asia_exact_for_comparison"] = bn.query("Lung cancer", event={"Smoker": True, "Dispnea": True}).to_numpy().tolist()

# ---- counting (fit, Chow-Liu) on 1M rows x 100 columns of the grid network's forward samples
from sorobn_amd import learning  # noqa: E402
codes = grid.backend.engine.sample(1_000_000, seed=5)           # uint8 codes in variable-id order
names = list(grid.backend.flat.names)
card = grid.backend.flat.card
ce = learning.counting_engine()
tabs = [tuple(grid.backend.flat.scope[v]) for v in range(len(names))]
ce.count_tables(codes[:1000], card, tabs)
t0 = time.perf_counter()
cnt = ce.count_tables(codes, card, tabs)
dt = time.perf_counter() - t0
out["8f_fit_counts_1M_rows_100_cpts"] = {"ms": dt * 1e3, "rows_per_s": 1e6 / dt, "cells_counted_per_s": 1e6 * len(tabs) / dt}
import itertools as _it  # noqa: E402
pairs = [(j,) for j in range(100)] + list(_it.combinations(range(100), 2))
t0 = time.perf_counter()
cnt = ce.count_tables(codes, card, pairs)
dt = time.perf_counter() - t0
out["8f_chow_liu_counts_1M_rows_4950_pairs"] = {"ms": dt * 1e3, "rows_per_s": 1e6 / dt, "cells_counted_per_s": 1e6 * len(pairs) / dt,
                                               "note": "includes the 100 MB host->device copy of the code matrix"}
Xs = pd.DataFrame({n: codes[:200_000, v] for v, n in enumerate(names)})
t0 = time.perf_counter()
tree = sorobn_amd.structure.chow_liu(Xs)
out["8f_chow_liu_200k_rows_100_columns_end_to_end"] = {"ms": (time.perf_counter() - t0) * 1e3, "edges": len(tree)}
learner = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet)
t0 = time.perf_counter()
learner.fit(Xs)
out["8f_fit_200k_rows_100_nodes_end_to_end"] = {"ms": (time.perf_counter() - t0) * 1e3,
                                               "max_abs_err_vs_true_cpt": float(max(np.max(np.abs((learner.P[n] - grid.P[n]).fillna(0).to_numpy())) for n in names))}
del codes, cnt

# ---- C5: Gibbs
spec5 = netspec.grid_spec(5, 10, 8, seed=0)
bn5 = netspec.build(spec5, sorobn_amd.BayesNet)
rng = np.random.default_rng(1)
ev = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
exact = bn5.query("025", event=ev)
chains, iters = 128, 100_000
bn5.query("025", event=ev, algorithm="gibbs", n_iterations=1000, n_chains=chains)
t0 = time.perf_counter()
got = bn5.query("025", event=ev, algorithm="gibbs", n_iterations=iters, n_chains=chains)
dt = time.perf_counter() - t0
out["C5_gibbs_128_chains_100k"] = {"wall_ms": dt * 1e3, "updates_per_s": chains * iters / dt,
                                   "max_abs_err_vs_exact": float(np.max(np.abs(got.to_numpy() - exact.to_numpy()))),
                                   "note": "one GPU's share (128 of 1024 chains); latency/LDS-bound, HBM roofline n/a"}
for chains in (1024, 8192):
    t0 = time.perf_counter()
    got = bn5.query("025", event=ev, algorithm="gibbs", n_iterations=iters, n_chains=chains)
    dt = time.perf_counter() - t0
    out[f"C5_gibbs_{chains}_chains_100k"] = {"wall_ms": dt * 1e3, "updates_per_s": chains * iters / dt,
                                             "max_abs_err_vs_exact": float(np.max(np.abs(got.to_numpy() - exact.to_numpy())))}
print(json.dumps(out, indent=1))
