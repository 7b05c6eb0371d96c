#!/usr/bin/env python
"""Benchmark of the exact-inference hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: `python bench.py --gpus N` spawns the N ranks itself (one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE in the
environment, the RCCL id through a private directory) - no PyTorch anywhere; launched by `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` (the driver's contract) the ranks torchrun started are used as they are.

A "step" is one pass of the hot path over one batch of synthetic requests: `--batch` exact posterior queries per GPU
(1 query node + 4 evidence nodes, the BASELINE C3 stream from default_rng(1)) on the synthetic 10x10 grid BN with 4
states per node (Dirichlet(1) CPTs from default_rng(0)).  Inputs (the flattened network) are resident in HBM before
the timed region; the timed region covers planning, program upload, the VE kernel, result download and - for N > 1 -
the RCCL all-gather of the posteriors.  Multi-GPU: one process per GPU, requests are independent so every rank
processes its own contiguous shard of the stream (weak scaling, no data-path collective besides the final gather).
The whole product path is ctypes -> libmibn.so (HIP kernels + RCCL through the C-ABI): PyTorch is never imported, and there
is no second transport behind mibn_comm_*: if it is unavailable on a rank, the launch exits non-zero.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  `roofline`      dominant kernel: algorithmic bytes per launch / HIP-event duration, vs the 8 TB/s HBM peak
  `cpu_baseline`  kind "reference": the UNMODIFIED reference (sorobn's pandas path, from oracle/_ref) timed on this box's
                  host cores on the first requests of the same stream, no cost filter, under a wall budget: one process
                  (cores 1 - the reference is single-threaded) and an N-process aggregate with N stated; the C port of
                  the oracle is reported next to it as `cpu_port`
  `configs`       measured in the same process after the C3 region: C1 (alarm, single query latency), C2 (Asia, 100 k batched queries),
                  C5 (Gibbs, 100 k updates x chains), each beside the reference where it runs; `C3_n_evidence_{1,8,16}`;
                  `C3_two_planner_threads`, `C3_planner_threads_{1,4}` (fresh engines with that many planning workers: the ranks of
                  an 8-GPU node with a small CPU quota) and `projected_8gpu` computed from them; `C3_query_many_pandas` (Python
                  request objects in, one pandas object out); `F1` - `F4`, the section-8(f) rows (predict_proba, likelihood
                  weighting, fit, Chow-Liu) with the unmodified reference timed beside each
`--config c5` times the Gibbs configuration instead (a step = 100 k single-site updates x 128 chains per GPU + the
int64 histogram reduce).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per launch come from rocprofv3 PMC passes of this same command (separate --pmc FETCH_SIZE / WRITE_SIZE runs,
# corrected as MI355X_MICROARCH.md prescribes and calibrated on known byte counts): they cannot be collected inside this
# process, and an absolute figure of another session next to this run's algorithmic bytes means nothing (other requests per
# launch) - so the line carries the RATIO traffic / algorithmic of the newest committed PMC session (profiles/r*_pmc.json,
# written by tools/make_pmc_json.py), both sides measured in that session, with its provenance; `traffic` itself is null.


def pmc_ratio(kernel):
    import glob
    best = None
    # (the newest session: by round, a round's closing sessions - "..._final_...", then "..._close<n>_..." - last)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")),
                    key=lambda p: (os.path.basename(p)[:3], 2 if "_close" in os.path.basename(p) else int("_final_" in p), os.path.basename(p))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        k = d.get("per_kernel", {}).get(kernel)
        if k is None:  # (the engine's kernel names are 47 characters at most: the level group's name arrives cut)
            k = next((v for name, v in d.get("per_kernel", {}).items() if len(kernel) >= 40 and name.startswith(kernel)), None)
        if k is None and kernel.startswith("level:"):  # (sessions before round 6 named the level's concurrent launches after two of its kernels)
            k = d.get("per_kernel", {}).get("ve_level_kernel||ve_sweep_dma_kernel")
        if k and k.get("alg_bytes_per_launch_same_run"):
            best = (f, k)
    if not best:
        return None, None
    f, k = best
    return k["traffic_bytes_per_launch"] / k["alg_bytes_per_launch_same_run"], {
        "file": os.path.relpath(f, ROOT), "session_traffic_bytes_per_launch": k["traffic_bytes_per_launch"],
        "session_alg_bytes_per_launch": k["alg_bytes_per_launch_same_run"], "fetch_correction": k.get("fetch_correction")}


# ------------------------------------------------------------------------------------------------ CPU baselines

def _run_ref_workers(specs, wall_s):
    """Spawn oracle/ref_worker.py once per entry of `specs` (argument lists), collect their JSON lines.  -> list of
    (closing record with "times" and "answers", stderr tail) or None per worker that failed."""
    env = dict(os.environ, PYTHONHASHSEED="0", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    worker = os.path.join(ROOT, "oracle", "ref_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, *map(str, sp)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for sp in specs]
    out = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=wall_s + 120)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        done, times, answers = None, {}, {}
        for line in so.splitlines():
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if d.get("done"):
                done = d
            elif "i" in d:
                answers[d["i"]] = (d["index"], d["values"])
                times[d["i"]] = d["s"]
        if done is not None:
            done["times"], done["answers"] = times, answers
        out.append((done, se[-400:]))
    return out


def cpu_reference(cap_s, n_req=16):
    """The unmodified reference on this box's host cores (oracle/ref_worker.py), on a FIXED request set: requests 0 .. n_req - 1
    of the C3 stream, one single-threaded process each (the reference cannot use more than one core), every request under
    the same cap of `cap_s` seconds of wall time; a request that does not finish is abandoned and its cap counted.
      single    the per-core rate: finished / sum of the per-request times - what ONE core does on this request set
      aggregate finished / wall time of the n_req processes running side by side (cores = n_req)
    Round 2 ran "whatever fits a 20 s budget" - three requests, +-50 % from run to run; the same sixteen requests every time
    make the figure comparable between runs and boxes.  Returns (single, aggregate, answers) or None without oracle/_ref."""
    from oracle import refload
    if not refload.available():
        return None
    res = _run_ref_workers([["--workload", "c3", "--first", n_req, "--shard", i, "--nshards", n_req, "--budget", cap_s] for i in range(n_req)], cap_s)
    if any(d is None for d, _ in res):
        bad = next(se for d, se in res if d is None)
        return {"error": f"reference worker failed: {bad}"}, None, {}
    answers, secs = {}, []
    for i, (d, _) in enumerate(res):
        answers.update(d["answers"])
        secs.append(d["times"].get(i))  # None: abandoned at the cap
    fin = sum(t is not None for t in secs)
    spent = sum(t if t is not None else cap_s for t in secs)
    wall = max(d["elapsed"] for d, _ in res)
    dist = ", ".join(f"{t:.1f}" if t is not None else f">{cap_s:.0f}" for t in secs)
    single = {"value": fin / spent if spent > 0 else 0.0, "unit": "queries/s", "cores": 1, "kind": "reference",
              "sample": f"unmodified sorobn (oracle/_ref, loaded from '{res[0][0]['reference']}'; BayesNet.query, pandas 2.3.3, PYTHONHASHSEED=0, "
                        f"hash-ordered names => row-major elimination) on the FIXED request set 0..{n_req - 1} of the C3 stream (rng seed 1), "
                        f"NO cost filter, one single-threaded process per request, cap {cap_s:.0f} s each: {fin} finished, {n_req - fin} abandoned "
                        f"(cap counted); value = finished / sum of per-request seconds = the rate of ONE core; seconds per request: [{dist}]",
              "finished": fin, "attempted": n_req, "core_seconds": spent, "seconds_per_request": secs, "cap_s": cap_s}
    aggregate = {"value": fin / wall if wall > 0 else 0.0, "unit": "queries/s", "processes": n_req, "cores": n_req, "finished": fin,
                 "attempted": n_req, "elapsed_s": wall,
                 "sample": f"the same {n_req} processes side by side: finished / wall time ({n_req} cores; quota of this box: see host)"}
    return single, aggregate, answers


def cpu_reference_small(budget_s=40.0, n=2000):
    """C1 / C2 beside their GPU figures ("time all requests or a 2 000-request sample"): the reference's query() on
    the alarm request of config 1, repeated, and on the first `n` requests of the Asia stream of config 2 - one process each,
    side by side."""
    from oracle import refload
    if not refload.available():
        return {}
    res = _run_ref_workers([["--workload", w, "--first", n, "--shard", 0, "--nshards", 1, "--budget", budget_s] for w in ("c1", "c2")], budget_s)
    out = {}
    for w, (d, se) in zip(("c1", "c2"), res):
        if d is None:
            out[w] = {"error": se}
            continue
        out[w] = {"value": d["finished"] / d["elapsed"] if d["elapsed"] > 0 else 0.0, "unit": "queries/s", "cores": 1, "kind": "reference",
                  "ms_per_query": 1e3 * d["elapsed"] / max(1, d["finished"]),
                  "sample": f"unmodified sorobn (oracle/_ref): BayesNet.query on " +
                            ("the config-1 request, repeated" if w == "c1" else "the first requests of the config-2 Asia stream (seed 0)") +
                            f": {d['finished']} of {n} requests in {d['elapsed']:.1f} s (budget {budget_s:.0f} s), one core"}
    return out


def cpu_port(spec, qv, ev, ec, gpu_post, budget_s, max_rows=3e7):
    """The C oracle (oracle/ve_oracle.c, kind="port": sparse-table VE restating bayes_net.py:739-794, eliminating in
    ascending-name = row-major order like the hash-ordered reference) on the first requests of the stream until
    `budget_s` seconds are spent; also returns the max-abs marginal error of the GPU posteriors on that sample.
    Requests whose row-major product would exceed `max_rows` rows (minutes each on the CPU) are skipped and counted -
    so this second figure is optimistic; the unfiltered one is `cpu_baseline` (the reference itself)."""
    import netspec
    from oracle.oracle import OracleNet

    on = OracleNet(spec)
    oid = np.array([on.id[f"{i:03d}"] for i in range(len(on.names))], np.int32)  # grid id -> oracle id
    prio = np.empty(len(on.names), np.int32)
    prio[oid] = np.arange(len(on.names), dtype=np.int32)
    t0 = time.perf_counter()
    n = skipped = i = 0
    err = 0.0
    while i < len(qv) and time.perf_counter() - t0 < budget_s:
        if netspec.grid_row_major_cost(qv[i], ev[i], 10, 10, 4)[0] > max_rows:
            skipped += 1
            i += 1
            continue
        codes, vals = on.query_codes([int(oid[qv[i]])], oid[ev[i]].tolist(), ec[i].tolist(), order=prio)
        dense = np.zeros(int(on.card[int(oid[qv[i]])]))
        dense[codes[:, 0]] = vals
        err = max(err, float(np.max(np.abs(dense - gpu_post[i]))))
        n += 1
        i += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt if n else 0.0, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"{n} of the first {n + skipped} requests of the C3 stream in {dt:.1f} s; the other {skipped} "
                      f"(> {max_rows:.0e} row-major product rows) were NOT run, so this figure is an upper bound; "
                      "oracle/ve_oracle.c, single thread, row-major elimination order"}, err


# ------------------------------------------------------------------------------------------------ other configs

def other_configs(device, with_cpu=True):
    """C1 / C2 / C5 of BASELINE.json in this process (N = 1, after the C3 region)."""
    import golden_util as gu
    import netspec
    import sorobn_amd

    out = {}
    nets = {n["spec"]["name"]: n["spec"] for n in gu.load("examples.json")}
    # C1: alarm, one query() through the reference-shaped API
    bna = netspec.build(nets["alarm"], sorobn_amd.BayesNet).use_device(device)
    ev1 = {"Mary calls": True, "John calls": True}
    ans = bna.query("Burglary", event=ev1)
    per = []
    for _ in range(5):  # five batches of 200 blocking calls, the median batch (a single batch swings with the clocks of an idle GPU)
        t0 = time.perf_counter()
        for _ in range(200):
            ans = bna.query("Burglary", event=ev1)
        per.append((time.perf_counter() - t0) / 200 * 1e3)
    out["C1_alarm_single_query"] = {"ms_per_query": sorted(per)[2], "ms_per_query_batches": per, "answer": ans.to_numpy().tolist(),
                                    "reference_answer": [0.7158281646356071, 0.28417183536439294]}
    # C2: Asia, 100 k requests of the SURVEY 8(d) stream in one batch (host-side encode of the names outside the timing)
    bn2 = netspec.build(nets["asia"], sorobn_amd.BayesNet).use_device(device)
    be2 = bn2.backend
    eng2 = be2.engine
    reqs = netspec.asia_requests(list(bn2.nodes), 100_000, seed=0)
    q_off = np.arange(len(reqs) + 1, dtype=np.int64)
    q_vars = np.array([be2.flat.id[q] for q, _ in reqs], np.int32)
    e_off = np.concatenate([[0], np.cumsum([len(e) for _, e in reqs])]).astype(np.int64)
    e_vars = np.array([be2.flat.id[k] for _, e in reqs for k in e], np.int32)
    e_codes = np.array([be2.flat.code_of(be2.flat.id[k], v) for _, e in reqs for k, v in e.items()], np.int32)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        post, off = eng2.query_batch(q_off, q_vars, e_off, e_vars, e_codes)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    s = eng2.stats()
    sums = np.add.reduceat(post, off[:-1])
    nz = int((sums > 0).sum())
    out["C2_asia_100k"] = {"queries_per_s": nz / best, "wall_ms": best * 1e3, "kernel_ms": s["kernel_ms"], "plan_ms": s["plan_ms"],
                           "launches": s["n_launches"], "zero_probability_evidence_excluded": int(len(reqs) - nz),
                           "note": "36 CPT numbers: not HBM-bound (roofline n/a); host- and launch-bound"}
    # C5: 50-node K=8 grid, Gibbs, 100 k single-site updates per chain; 128 chains = one GPU's share of the 1024
    spec5 = netspec.grid_spec(5, 10, 8, seed=0)
    bn5 = netspec.build(spec5, sorobn_amd.BayesNet).use_device(device)
    rng = np.random.default_rng(1)
    ev5 = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
    exact = bn5.query("025", event=ev5).to_numpy()
    bn5.query("025", event=ev5, algorithm="gibbs", n_iterations=1000, n_chains=128)
    for chains in (128, 1024):
        t0 = time.perf_counter()
        got = bn5.query("025", event=ev5, algorithm="gibbs", n_iterations=100_000, n_chains=chains).to_numpy()
        dt = time.perf_counter() - t0
        out[f"C5_gibbs_{chains}_chains_x_100k"] = {"wall_ms": dt * 1e3, "updates_per_s": chains * 100_000 / dt,
                                                   "max_abs_err_vs_exact": float(np.max(np.abs(got - exact)))}
    out["C5_note"] = "latency/LDS-bound (CPTs resident in LDS), HBM roofline n/a; 128 chains = one GPU's share of config 5"
    if with_cpu:
        small = cpu_reference_small()
        if "c1" in small:
            out["C1_alarm_single_query"]["cpu_baseline"] = small["c1"]
        if "c2" in small:
            out["C2_asia_100k"]["cpu_baseline"] = small["c2"]
    return out


def f_configs(device, with_cpu=True):
    """SURVEY section 8(f) rows in the line: predict_proba, likelihood weighting, fit, Chow-Liu - each on the
    GPU path and, side by side on this box's host (one single-threaded process per row: oracle/ref_worker.py --workload f1..f4),
    on the UNMODIFIED reference with the same inputs.  None of them is HBM-bound; the bound is named per row."""
    import netspec
    import pandas as pd
    import sorobn_amd

    frame = lambda rows, names, K, seed: pd.DataFrame(np.random.default_rng(seed).integers(0, K, (rows, len(names))), columns=names)
    sizes = {"f1": 200_000, "f2": 300_000, "f3": 1_000_000, "f4": 20_000}
    ref = {}
    procs = None
    if with_cpu:
        from oracle import refload
        if refload.available():
            env = dict(os.environ, PYTHONHASHSEED="0", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            worker = os.path.join(ROOT, "oracle", "ref_worker.py")
            procs = {w: subprocess.Popen([sys.executable, worker, "--workload", w, "--rows", str(n), "--budget", "60"], env=env,
                                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w, n in sizes.items()}
    out = {}

    def best_of(fn, reps=2):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, r

    # F1 predict_proba (bayes_net.py:934-962 over full_joint_dist 398-465): a 3x3 K=4 grid - the largest grid whose 4^9-row full
    # joint the reference builds in reasonable time - all nine columns observed; and the 10x10 grid with three observed columns,
    # which the reference cannot answer at all (a 4^100-row joint): there the 97 other variables are eliminated on the device
    spec33 = netspec.grid_spec(3, 3, 4, seed=0)
    bn33 = netspec.build(spec33, sorobn_amd.BayesNet).use_device(device)
    X1 = frame(sizes["f1"], list(spec33["nodes"]), 4, 11)
    bn33.predict_proba(X1.iloc[:8])
    dt, pp = best_of(lambda: bn33.predict_proba(X1))
    out["F1_predict_proba_grid"] = {"rows_per_s": len(X1) / dt, "rows": len(X1), "seconds": dt, "network": "3x3 grid, K=4, nine observed columns",
                                    "checksum": float(pp.sum()), "bound": "host: the rows' labels encoded column by column (Index.get_indexer) + one gather from the dense joint (the device "
                                    "computes the 262 144-cell joint once, 2 MB: HBM roofline n/a); the reference looks the rows up in a pandas MultiIndex"}
    grid = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet).use_device(device)
    Xg = frame(100_000, ["011", "055", "090"], 4, 1)
    grid.predict_proba(Xg.iloc[:4])
    dt, _ = best_of(lambda: grid.predict_proba(Xg))
    out["F1_predict_proba_grid10x10_3_columns"] = {"rows_per_s": len(Xg) / dt, "rows": len(Xg), "seconds": dt,
                                                   "reference": "cannot run: predict_proba builds the full joint (4^100 rows)",
                                                   "bound": "host label encoding + gather; the elimination of the 97 unobserved variables is one exact query"}
    # F2 likelihood weighting (bayes_net.py:621-663, forward sampling 518-575) on Asia
    asia = netspec.build({n["spec"]["name"]: n["spec"] for n in __import__("golden_util").load("examples.json")}["asia"], sorobn_amd.BayesNet).use_device(device)
    ev2 = {"Smoker": True, "Dispnea": True}
    asia.query("Lung cancer", event=ev2, algorithm="likelihood", n_iterations=1000)
    n2 = 16_000_000
    dt, a2 = best_of(lambda: asia.query("Lung cancer", event=ev2, algorithm="likelihood", n_iterations=n2))
    out["F2_likelihood_weighting"] = {"samples_per_s": n2 / dt, "samples": n2, "seconds": dt, "answer": a2.to_numpy().tolist(),
                                      "exact": asia.query("Lung cancer", event=ev2).to_numpy().tolist(),
                                      "bound": "latency / Philox throughput: one sample per lane walks 8 variables, CPTs in LDS; HBM roofline n/a"}
    # F3 fit (bayes_net.py:467-516): 1 M rows x 100 four-state columns onto the 10x10 grid's structure, end to end (factorise the
    # label columns on the host, one counting launch for the 100 contingency tables, the CPT Series)
    X3 = frame(sizes["f3"], [f"{i:03d}" for i in range(100)], 4, 12)
    learner = netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet).use_device(device)
    learner.fit(X3.iloc[:1000])
    dt, _ = best_of(lambda: learner.fit(X3))
    out["F3_fit_grid"] = {"rows_per_s": len(X3) / dt, "rows": len(X3), "columns": 100, "seconds": dt,
                          "bound": "host factorisation of 100 label columns + the PCIe copy of the 100 MB code matrix; the count kernel itself is LDS-atomic bound"}
    # F4 structure.chow_liu (structure.py:9-63): 100 four-state columns, all 4 950 pairwise tables in one counting launch
    X4 = frame(sizes["f4"], [f"{i:03d}" for i in range(100)], 4, 13)
    sorobn_amd.structure.chow_liu(X4.iloc[:500])
    dt, tree = best_of(lambda: sorobn_amd.structure.chow_liu(X4))
    X4b = frame(200_000, [f"{i:03d}" for i in range(100)], 4, 13)
    dtb, _ = best_of(lambda: sorobn_amd.structure.chow_liu(X4b), reps=1)
    out["F4_chow_liu_100cols"] = {"seconds": dt, "rows": len(X4), "rows_per_s": len(X4) / dt, "edges": len(tree),
                                  "seconds_200k_rows": dtb, "bound": "LDS atomics of the count kernel (4 950 16-cell tables per row block) + the host's "
                                  "mutual-information arithmetic and Kruskal over 4 950 edges"}
    from sorobn_amd import learning
    for b in (bn33, grid, asia, learner):
        b.backend.engine.close()
    for e in list(learning._engines.values()):
        e.close()
    learning._engines.clear()
    if procs:
        key = {"f1": ("F1_predict_proba_grid", "rows_per_s"), "f2": ("F2_likelihood_weighting", "samples_per_s"), "f3": ("F3_fit_grid", "rows_per_s"),
               "f4": ("F4_chow_liu_100cols", "rows_per_s")}
        for w, pr in procs.items():
            try:
                so, se = pr.communicate(timeout=200)
            except subprocess.TimeoutExpired:
                pr.kill()
                so, se = pr.communicate()
            d = None
            for line in so.splitlines():
                try:
                    j = json.loads(line)
                except ValueError:
                    continue
                if j.get("done"):
                    d = j
            name, unit = key[w]
            if d is None:
                out[name]["cpu_baseline"] = {"error": se[-300:]}
                continue
            rate = d["units"] / d["elapsed"] if d["finished"] else 0.0
            out[name]["cpu_baseline"] = {"value": rate, "unit": unit.replace("_per_s", "/s"), "cores": 1, "kind": "reference", "seconds": d["elapsed"],
                                         "finished": bool(d["finished"]),
                                         "sample": f"unmodified sorobn (oracle/_ref, '{d['reference']}') on the same input: {d['units']} "
                                                   f"{'samples' if w == 'f2' else 'rows'}, one single-threaded process, beside the three other 8(f) "
                                                   "reference processes" + ("; its sampler is oracle/refload's pure-Python stand-in for the absent third-party "
                                                                            "`vose` (Cython in the original): the reference's own rate would be higher" if w == "f2" else "")}
            if rate > 0:
                ours = out[name][unit] if w != "f2" else out[name]["samples_per_s"]
                out[name]["speedup_vs_reference_one_core"] = ours / rate
    return out


def c3_pandas(bn, n=131_072, sub_batch=32768):
    """throughput THROUGH the drop-in pandas boundary.  The C3 stream as a Python user holds it - a list of
    (query tuple, event dict) with node names and labels - into `BayesNet.query_many` (validation, bulk encode of names and labels,
    sub-batches with two engine calls in flight) and out as ONE pandas object with every posterior (`PosteriorBatch.to_frame()`);
    building the request list is outside the clock (it is the caller's data), everything else inside.  Beside it: the cost of
    materialising individual Series (`batch[i]`, identical to `query()`'s - checked on a sample with pandas' strict comparison)."""
    import netspec
    import pandas as pd
    q, ev, ec = netspec.c3_requests(100, 4, 2 * n, 4, seed=1)
    mk = lambda lo, hi: [((f"{a:03d}",), {f"{v:03d}": int(c) for v, c in zip(vs, cs)})
                         for a, vs, cs in zip(q[lo:hi].tolist(), ev[lo:hi].tolist(), ec[lo:hi].tolist())]
    warm, reqs = mk(0, n), mk(n, 2 * n)
    bn.query_many(warm, sub_batch=sub_batch).to_frame()
    t0 = time.perf_counter()
    batch = bn.query_many(reqs, sub_batch=sub_batch)
    t1 = time.perf_counter()
    frame = batch.to_frame()
    t2 = time.perf_counter()
    k = 2000
    series = [batch[i] for i in range(k)]
    t3 = time.perf_counter()
    for i in (0, 1, 17, k - 1):
        pd.testing.assert_series_equal(series[i], bn.query(*reqs[i][0], event=reqs[i][1]), check_exact=True)
    t4 = time.perf_counter()
    for i in range(200):
        bn.query(*reqs[i][0], event=reqs[i][1])
    t5 = time.perf_counter()
    return {"queries_per_s": n / (t2 - t0), "requests": n, "seconds": t2 - t0, "query_many_s": t1 - t0, "to_frame_s": t2 - t1,
            "frame_rows": int(len(frame)), "series_us_each": (t3 - t2) / k * 1e6,
            "single_query_ms_on_this_network": (t5 - t4) / 200 * 1e3,
            "series_identical_to_query": True,
            "note": "requests in as Python (tuple, dict) objects with names and labels, answers out as one pandas object (a DataFrame: request, "
                    "variables, cell, p - the query variable changes from request to request); batch[i] builds the Series query() returns"}


def c3_variant(eng, to_var, n_evidence, calls=12, warmup_calls=10, batch=32768):
    """A short stepped run of the C3 stream with another number of evidence nodes, or on another engine ("also report the
    n_evidence in {1, 8, 16} variants"; VERDICT r3: the device-planned 2-thread rank): `calls` pipelined engine calls of `batch`
    requests after `warmup_calls` (ten: the adaptive policy judges windows of two to three calls and moves the device's share of the planning in
    steps - round 5's session C measured 421 k queries/s for n_evidence = 16 two calls after the switch and 518 k with the share it
    converges to; the timed region starts with an idle GPU, so the planning of its first call is not hidden: twelve calls
    and more keep that ramp below a tenth), N = 1.  -> queries/s, MB per query, all-kernels GB/s, planner wall vs GPU busy time."""
    import netspec
    from sorobn_amd import sharding
    n = (calls + warmup_calls) * batch
    q, ev, ec = netspec.c3_requests(100, 4, n, n_evidence, seed=1)
    stream = sharding.ShardedStream(eng, sharding.SoloComm(), to_var[q][:, None], to_var[ev], ec, batch, sub_batch=batch)
    stream.run(range(warmup_calls))
    eng.drain()
    s0, k0 = eng.total_stats(), eng.total_kernel_stats()
    t0 = time.perf_counter()
    res = stream.run(range(warmup_calls, warmup_calls + calls))
    eng.drain()
    dt = time.perf_counter() - t0
    s1, k1 = eng.total_stats(), eng.total_kernel_stats()
    d = {k: s1[k] - s0[k] for k in s1}
    planned = k1.get("order_kernel+emit_kernel", {}).get("items", 0.0) - k0.get("order_kernel+emit_kernel", {}).get("items", 0.0)
    planner_ms = k1.get("order_kernel+emit_kernel", {}).get("ms", 0.0) - k0.get("order_kernel+emit_kernel", {}).get("ms", 0.0)
    nq = calls * batch
    assert res["requests"] == nq and abs(res["mass"] - nq) < 1e-6 * nq
    return {"queries_per_s": nq / dt, "requests": nq, "seconds": dt, "alg_MB_per_query": d["alg_bytes"] / nq / 1e6,
            "all_kernels_GBps": d["alg_bytes"] / max(1e-9, d["kernel_ms"]) / 1e6,
            "frac_of_hbm_peak": d["alg_bytes"] / max(1e-9, d["kernel_ms"]) / 1e6 / HBM_PEAK_GBS,
            "gpu_busy_ms_per_call": d["kernel_ms"] / calls, "planner_wall_ms_per_call": d["plan_ms"] / calls, "wall_ms_per_call": dt / calls * 1e3,
            "gpu_bound": d["kernel_ms"] >= 0.9 * dt * 1e3, "device_planned_requests_per_call": planned / calls,
            # the device planner's kernel is GPU time too (wave_plan_kernel takes the chip while it runs: the VE kernels' busy time - `gpu_busy`,
            # what `gpu_bound` compares with the wall time - does not contain it; HIP events around the planner's launches)
            "device_planner_ms_per_call": planner_ms / calls,
            "gpu_bound_counting_the_planner": d["kernel_ms"] + planner_ms >= 0.9 * dt * 1e3}


# ------------------------------------------------------------------------------------------------ transports

def make_comm(backend, world, rank, local_rank, engine):
    """-> (comm, transport name).  "rccl" = the C-ABI's mibn_comm_* (RCCL over xGMI, no PyTorch) - the only transport of a
    measurement: whatever goes wrong between dlopen and ncclCommInitRank raises, the rank exits non-zero and the launcher ends
    the launch (a fallback would mask a real RCCL defect in the first N > 1 line).  "files" = the dry run."""
    from sorobn_amd import sharding
    if world == 1:
        return sharding.SoloComm(), "none"
    if backend == "files":  # the dry run: everything of the N > 1 path but ncclCommInitRank and the collectives (sharding.FileComm)
        return sharding.FileComm(engine, rank, world), "files"
    if backend != "rccl":
        raise SystemExit(f"MIBN_BENCH_BACKEND={backend!r}: only 'rccl' (default) and 'files' (dry run) exist")
    return sharding.RcclComm(engine, rank, world), "rccl"


def pci_numbers(info):
    """The PCI bus id in mibn_device_info's line ("... pci 0000:05:00.0 ...") as [domain, bus, device, function] floats (what the
    ranks all-gather for the line's per_rank.device_pci), [-1] * 4 if it cannot be read."""
    import re
    m = re.search(r"pci ([0-9a-fA-F]+):([0-9a-fA-F]+):([0-9a-fA-F]+)\.([0-9a-fA-F]+)", info or "")
    return [float(int(g, 16)) for g in m.groups()] if m else [-1.0] * 4


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) with the environment a launcher
    would give them - RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR / MASTER_PORT (unused here: no socket is opened on it) - plus a private directory and a nonce for the out-of-band RCCL id (sharding.exchange_id), wait for them,
    and pass rank 0's line through.  No PyTorch: the ranks talk RCCL through the C-ABI (mibn_comm_*)."""
    import secrets
    import shutil
    import socket
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    comm_dir = tempfile.mkdtemp(prefix="mibn_launch_")
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                MIBN_COMM_DIR=comm_dir, MIBN_LAUNCH_NONCE=secrets.token_hex(8), MIBN_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=dict(base, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(n)]
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:  # a rank died: the others would wait for it in a collective for ever
                    rc = code
                    for other in pending:
                        other.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(comm_dir, ignore_errors=True)
    return rc


def host_cpu_quota():
    """CPUs' worth of time this container may use: the cgroup quota (v2 cpu.max / v1 cfs_quota), else the hardware threads."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    return float(os.cpu_count() or 1)


def project_8gpu(out):
    """What this box's measurements say about eight ranks on one node (no 8-GPU node is available to the builder; the driver's
    SCALE run is the measurement, this is the arithmetic behind DESIGN section 7, reproducible from the line): a rank of an 8-GPU
    node whose container has the SAME CPU quota as this box plans with quota / 8 threads; its rate is read off the measured
    few-threads configs (linear interpolation between the measured thread counts, the headline = the whole quota), the node's
    rate is 8 x that (requests are independent, the gather moves 32 B per query)."""
    quota = host_cpu_quota()
    pts = {}
    for nt, name in ((1, "C3_planner_threads_1"), (2, "C3_two_planner_threads"), (4, "C3_planner_threads_4")):
        v = out.get("configs", {}).get(name, {}).get("queries_per_s")
        if v:
            pts[float(nt)] = float(v)
    pts[max(quota, 5.0)] = float(out["value"])  # the headline: this box's whole quota on one rank
    xs = sorted(pts)
    t = quota / 8.0
    if t <= xs[0]:
        rate = pts[xs[0]] * min(1.0, t / xs[0]) if t < 1.0 else pts[xs[0]]
    elif t >= xs[-1]:
        rate = pts[xs[-1]]
    else:
        hi = next(x for x in xs if x >= t)
        lo = max(x for x in xs if x <= t)
        rate = pts[lo] if hi == lo else pts[lo] + (pts[hi] - pts[lo]) * (t - lo) / (hi - lo)
    return {"host_cpu_quota": quota, "planner_threads_per_rank_on_8_gpus_same_quota": t,
            "measured_queries_per_s_by_planner_threads": {str(int(x)) if x == int(x) else str(x): pts[x] for x in xs},
            "per_rank_queries_per_s": rate, "node_queries_per_s": 8.0 * rate, "scaling_vs_this_line": 8.0 * rate / float(out["value"]),
            "scaling_if_each_rank_keeps_this_quota": 8.0,
            "note": "PROJECTION from one-GPU measurements of this run, not a measurement: 8 x the rate of a rank restricted to (this box's CPU "
                    "quota / 8) planning threads (adaptive policy: the rank's own GPU plans what its host threads cannot); the all-gather of "
                    "32 B per query over xGMI is not modelled (2^18 x 32 B = 8 MB per step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c3", "c5"])
    ap.add_argument("--batch", type=int, default=0, help="requests per engine call (mibn_submit_batch) = per chunk of the engine; 0 (default): a rank's shard of a step in "
                    "ceil(shard / --call-cap) equal calls (N = 1: 5 x 52 429, N = 2: 3 x 43 691, N = 4: 2 x 32 768, N = 8: 1 x 32 768); with --scaling weak "
                    "also the requests per step and GPU (then 32 768 unless given)")
    ap.add_argument("--call-cap", type=int, default=52_429, help="most requests of an engine call when --batch is 0: a C3 chunk of this size needs 224 GB of "
                    "arena, inside the engine's budget of 0.8 x the free HBM (profiles/r04_m_chunk.log: +1.5 %% queries/s over 32 768)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, BASELINE config 4): a step = --global-batch requests of the stream whatever N, split into "
                         "contiguous shards over the N ranks; weak: a step = N x --batch requests")
    ap.add_argument("--global-batch", type=int, default=262_144, help="requests per step with --scaling strong: 2^18, a quarter of a 2^20-request stream - "
                    "whole 32 768-request engine calls on 1, 2, 4 and 8 ranks (round 4's first sessions used 250 000: a tail call of 20 624 per step)")
    ap.add_argument("--n-evidence", type=int, default=4)
    ap.add_argument("--balance", default="count", choices=["count", "cost"],
                    help="split of a step's global batch over the ranks: equal counts, or equal planner cost estimates")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="cap per request of the reference leg of cpu_baseline (wall seconds)")
    ap.add_argument("--cpu-procs", type=int, default=16, help="size of the fixed request set of the reference leg (one process per request)")
    ap.add_argument("--full-stream", action="store_true",
                    help="after the stepped measurement: BASELINE config 3 / 4 as written - the first 1 M requests of the stream once, end to "
                         "end, contiguous shards over the N ranks, ONE all-gather (also with N > 1)")
    ap.add_argument("--port-seconds", type=float, default=6.0, help="wall budget of the C-port leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
    ap.add_argument("--only-configs", action="store_true", help="measure C1 / C2 / C5 and the section-8(f) rows only and print them (quick check)")
    ap.add_argument("--no-adaptive", action="store_true", help="keep the planner's full order search even when the host is the bottleneck")
    ap.add_argument("--sync", action="store_true", help="one blocking mibn_query_batch per step instead of the two-deep pipeline")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (experiments), e.g. --opt chunk=32768")
    ap.add_argument("--threads", type=int, default=0, help="planner threads of this rank (0 = host threads / ranks on the node)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and not os.environ.get("MIBN_BENCH_CHILD"):
            sys.exit(spawn_ranks(a.gpus))  # plain `python bench.py --gpus N`: this process becomes the launcher
        a.gpus = world
    if a.batch <= 0:
        if a.scaling == "strong":
            shard = -(-a.global_batch // world)
            a.batch = -(-shard // max(1, -(-shard // max(1024, a.call_cap))))
        else:
            a.batch = 32768

    # Transport of the final gather: "rccl" (default) = mibn_comm_* of the C-ABI, RCCL over xGMI, no PyTorch - and nothing behind
    # it.  MIBN_BENCH_BACKEND=files is the dry run: it lets the N > 1 path run on a box with fewer GPUs than ranks (the ranks then
    # share devices and the collectives go through files of the launch's directory, sharding.FileComm).
    backend = os.environ.get("MIBN_BENCH_BACKEND", "rccl")
    from sorobn_amd import _capi
    n_dev = max(1, _capi.device_count())
    device = local_rank if backend != "files" else local_rank % n_dev

    import netspec
    import sorobn_amd
    from sorobn_amd import sharding

    if a.only_configs:
        cfg = other_configs(device, with_cpu=not a.no_cpu)
        cfg.update(f_configs(device, with_cpu=not a.no_cpu))
        print(json.dumps(cfg), flush=True)
        return
    if a.config == "c5":
        return run_c5(a, rank, world, local_rank, device, backend)

    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn_amd.BayesNet).use_device(device)
    be = bn.backend  # flatten + upload: network resident in HBM from here on
    eng = be.engine
    if a.threads:
        eng.set_option("threads", a.threads)
    if not a.no_adaptive:
        # planning effort follows the host: with few cores per GPU (8 ranks on a small CPU quota) the min-fill search
        # is reserved for the expensive requests; never triggers while planning hides under the kernels (N = 1 here)
        eng.set_option("adaptive", 1)
    if a.batch > 32768:  # one chunk per engine call: the chunk size and the arena budget follow the call size (the engine caps the budget at 0.8 x free HBM)
        eng.set_option("chunk", a.batch)
        eng.set_option("arena_gb", 250)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, float(v))
    comm, transport = make_comm(backend, world, rank, local_rank, eng)

    total_steps = a.warmup + a.steps
    G = a.global_batch if a.scaling == "strong" else world * a.batch  # a step's global batch
    n_req = total_steps * G
    qv, ev, ec = netspec.c3_requests(100, 4, n_req, a.n_evidence, seed=1)
    # names "000".."099" sort like the ids, but variable ids follow bn.nodes: map stream ids -> var ids
    to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)

    def barrier():
        comm.barrier()
        eng.synchronize()

    # A step = one pass of the hot path over one global batch: every rank works through its contiguous shard in calls of
    # --batch requests, two calls in flight (mibn_submit_batch / mibn_wait: the host plans call k + 1 while the GPU runs k, as a
    # server streaming batches would), and the step ends with ONE all-gather of the posteriors over xGMI (sharding.ShardedStream).
    # Every step is complete - posteriors on the host and, for N > 1, gathered on every rank - before the closing barrier.
    stream = sharding.ShardedStream(eng, comm, to_var[qv][:, None], to_var[ev], ec, G, sub_batch=a.batch, balance=a.balance,
                                    pipelined=not a.sync)

    def run(steps):
        res = stream.run(steps)
        eng.drain()  # every launch finished and its HIP-event time booked
        return res

    run(range(a.warmup))
    barrier()
    st0, ks0 = eng.total_stats(), eng.total_kernel_stats()
    t0 = time.perf_counter()
    timed = run(range(a.warmup, total_steps))
    barrier()
    dt = time.perf_counter() - t0
    st1, ks1 = eng.total_stats(), eng.total_kernel_stats()
    agg = {k: st1[k] - st0[k] for k in st1}
    kagg = {n: {f: ks1[n][f] - ks0.get(n, {}).get(f, 0.0) for f in ks1[n]} for n in ks1}
    kagg = {n: d for n, d in kagg.items() if d["launches"] > 0}
    # per rank: what it processed in the timed region (requests, section-8(d) bytes, GPU busy time) - the shard imbalance as run
    try:
        my_info = eng.device_info()
    except Exception as e:  # noqa: BLE001
        my_info = f"unavailable ({e!r})"
    per_rank = comm.allgather(np.array([[float(stream.ranges(a.warmup)[rank][1] - stream.ranges(a.warmup)[rank][0]), agg["alg_bytes"],
                                         agg["kernel_ms"], agg["plan_ms"], float(device), *pci_numbers(my_info),
                                         float(getattr(comm, "rccl_ranks", 0)), float(getattr(comm, "rccl_rank", -1))]]))[:, 0, :] if world > 1 else None
    if world > 1:
        dt = float(comm.allreduce_max([dt])[0])

    full = None
    if a.full_stream:
        # BASELINE config 3 (N = 1) / config 4 (N > 1) as written: the first 1 M requests of the stream, once, end to end - contiguous
        # shards over the ranks, every rank in pipelined calls of --batch requests, ONE all-gather at the end.  Request generation
        # outside the clock, everything else (with --balance cost: the cost estimates too) inside.
        n_full = 1_000_000
        fq, fe, fc = netspec.c3_requests(100, 4, n_full, a.n_evidence, seed=1)
        fs = sharding.ShardedStream(eng, comm, to_var[fq][:, None], to_var[fe], fc, n_full, sub_batch=a.batch, balance=a.balance,
                                    pipelined=not a.sync)
        barrier()
        f0 = eng.total_stats()
        t_full = time.perf_counter()
        fres = fs.run([0])
        eng.drain()
        barrier()
        dt_full = time.perf_counter() - t_full
        f1 = eng.total_stats()
        mine = np.array([[float(fs.shard_requests[rank]), f1["alg_bytes"] - f0["alg_bytes"], f1["kernel_ms"] - f0["kernel_ms"]]])
        ranks = comm.allgather(mine)[:, 0, :] if world > 1 else mine
        if world > 1:
            dt_full = float(comm.allreduce_max([dt_full])[0])
        full = {"requests": fres["requests"], "seconds": dt_full, "queries_per_s": fres["requests"] / dt_full, "n_gpus": world,
                "scaling": "strong", "shard_balance": a.balance, "gathers": 1 if world > 1 else 0,
                "posterior_mass": fres["mass"],  # = requests (every posterior sums to 1): nothing was skipped
                "per_rank_requests": ranks[:, 0].tolist(), "per_rank_alg_GB": (ranks[:, 1] / 1e9).tolist(),
                "per_rank_gpu_busy_ms": ranks[:, 2].tolist(),
                "imbalance_alg_bytes_max_over_mean": float(ranks[:, 1].max() / ranks[:, 1].mean()),
                "imbalance_gpu_busy_max_over_mean": float(ranks[:, 2].max() / max(1e-9, ranks[:, 2].mean())),
                "note": "the first 1 M requests of the C3 stream (rng seed 1) in one pass, beside the stepped figure `value`"}

    if rank == 0:
        n_queries = a.steps * G
        assert timed["requests"] == n_queries and abs(timed["mass"] - n_queries) < 1e-6 * n_queries, (timed["requests"], timed["mass"])
        my_requests = n_queries if world == 1 else a.steps * (stream.ranges(a.warmup)[0][1] - stream.ranges(a.warmup)[0][0])  # (rank 0's own)
        # dominant kernel = the specialisation with the most HIP-event time over the timed region
        # (the device planner's pair of kernels moves no section-8(d) bytes: it is listed under "kernels", never the roofline's kernel)
        dom = max((n for n in kagg if kagg[n]["alg_bytes"] > 0), key=lambda n: kagg[n]["ms"])
        launches = max(1.0, kagg[dom]["launches"])
        bytes_per_launch = kagg[dom]["alg_bytes"] / launches
        ms_per_launch = kagg[dom]["ms"] / launches
        achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
        all_kernels = agg["alg_bytes"] / (agg["kernel_ms"] * 1e-3) / 1e9
        out = {
            "metric": "exact posterior queries/sec on 100-node 4-state grid BN",
            "value": n_queries / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            # strong (default): a step is the same --global-batch requests of the stream whatever N, contiguous shards over the
            # ranks, one all-gather - BASELINE config 4; weak: every rank processes --batch requests of its own per step
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: 10x10 grid BN, 4 states/node, Dirichlet(1) CPTs rng(0); requests = "
                                   f"1 query + {a.n_evidence} evidence nodes, rng(1) stream",
                       "requests_per_step": G, "requests_per_step_per_gpu": G / world, "requests_per_engine_call": a.batch,
                       "parallelism": f"dp{world} (independent contiguous shards, one all-gather per step)",
                       "gather": {"none": "none", "rccl": "RCCL via the C-ABI (mibn_comm_allgather_f64), no PyTorch",
                                  "files": "DRY RUN (MIBN_BENCH_BACKEND=files): launch, librccl probe, vote and id exchange as with "
                                           "RCCL, the collectives through files - a check of the plumbing, not a measurement"}[transport],
                       # ncclCommCount of the live communicator (0: no RCCL communicator - N = 1 or the dry run)
                       "rccl_ranks": int(getattr(comm, "rccl_ranks", 0)),
                       "shard_balance": a.balance, "planner_threads": a.threads or "auto (cgroup quota / ranks)",
                       "adaptive_planning": not a.no_adaptive,
                       # chunks planned by order_kernel + emit_kernel in the timed region (option gpu_emit, or the adaptive policy when
                       # the host's planning workers bound the pipeline)
                       "device_planned_requests_per_step": kagg.get("order_kernel+emit_kernel", {}).get("items", 0.0) / a.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": dom, "alg_bytes_per_launch": bytes_per_launch,
                         "ms_per_launch": ms_per_launch, "launches": launches,
                         "share_of_kernel_time": kagg[dom]["ms"] / agg["kernel_ms"],
                         "all_kernels_GBps": all_kernels,
                         "alg_bytes_per_query": agg["alg_bytes"] / max(1, my_requests)},
            "kernels": {n: {"launches": d["launches"], "ms": d["ms"], "alg_GB": d["alg_bytes"] / 1e9,
                            "GBps": d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6}
                        for n, d in sorted(kagg.items(), key=lambda kv: -kv[1]["ms"])},
            # host clocks of a PIPELINED api, per step (rank 0) - they overlap and do not add up to ms_per_step: the planner works
            # under the previous call's kernels, the wait for a call's results contains its kernels
            "pipeline_clocks_ms_per_step": {"gpu_busy_ms": agg["kernel_ms"] / a.steps,
                                            "planner_wall_ms_inside_submit_calls": agg["plan_ms"] / a.steps,
                                            "upload_issue_ms": agg["h2d_ms"] / a.steps,
                                            "host_blocked_in_wait_ms": agg["d2h_ms"] / a.steps,
                                            "submit_calls_ms": agg["total_ms"] / a.steps,
                                            "gpu_bound": agg["kernel_ms"] >= 0.9 * (dt * 1e3)},
        }
        if per_rank is not None:
            out["per_rank"] = {"requests_per_step": per_rank[:, 0].tolist(), "alg_GB": (per_rank[:, 1] / 1e9).tolist(),
                               "gpu_busy_ms": per_rank[:, 2].tolist(), "planner_wall_ms": per_rank[:, 3].tolist(),
                               "device": [int(x) for x in per_rank[:, 4]],
                               "device_pci": ["%04x:%02x:%02x.%x" % tuple(int(x) for x in r[5:9]) if r[5] >= 0 else "?" for r in per_rank],
                               "rccl_comm_count": [int(x) for x in per_rank[:, 9]], "rccl_user_rank": [int(x) for x in per_rank[:, 10]],
                               "imbalance_alg_bytes_max_over_mean": float(per_rank[:, 1].max() / per_rank[:, 1].mean()),
                               "imbalance_gpu_busy_max_over_mean": float(per_rank[:, 2].max() / per_rank[:, 2].mean())}
        out["roofline"]["traffic_over_alg"], out["roofline"]["traffic_source"] = pmc_ratio(dom)
        # the exact path has two kernels since round 2 (ve_level_kernel, ve_sweep_kernel) with about the same share of the
        # time: the same figures for each of them, so that the line does not depend on which one is ahead in this run
        out["roofline"]["per_kernel"] = {}
        for n, d in kagg.items():
            if d["alg_bytes"] <= 0:  # (the device planner's pair of kernels: no section-8(d) bytes, see "kernels")
                continue
            la = max(1.0, d["launches"])
            gbps = d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6
            out["roofline"]["per_kernel"][n] = {"achieved": gbps, "frac": gbps / HBM_PEAK_GBS, "alg_bytes_per_launch": d["alg_bytes"] / la,
                                                "ms_per_launch": d["ms"] / la, "launches": la, "traffic": None, "traffic_over_alg": pmc_ratio(n)[0],
                                                "share_of_kernel_time": d["ms"] / agg["kernel_ms"]}
        if "||" in dom:
            out["roofline"]["note"] = ("option overlap (default): the launches of a level - one each of ve_level_kernel, ve_mfma_kernel, ve_sweep_dma_kernel and "
                                       "ve_segment_kernel, items of different requests - run CONCURRENTLY on four streams; the unit whose duration means anything "
                                       "is the level (from the earliest start to the latest end of its launches, HIP events): `kernel` names it, `launches` = levels; "
                                       "rocprofv3's kernel trace gives the same unit as the connected components of the kernels' intervals "
                                       "(tools/rocprof_summary.py, profiles/r04_o_rocprofv3_summary.txt).  The kernels' own event durations (per_kernel) then include "
                                       "each other's share of the chip; per_kernel_serialised = the same kernels right after the timed region with the launches "
                                       "serialised (overlap=0), for comparison with earlier rounds.")
        if world == 1 and not a.no_configs and "||" in dom:
            # the two kernels by themselves: a few calls with the launches of a level one after the other (option overlap = 0)
            try:
                eng.set_option("overlap", 0)
                k0 = eng.total_kernel_stats()
                stream.run(range(min(3, total_steps)))  # (three steps: VERDICT r4 weak 5 - one was thin)
                eng.drain()
                k1 = eng.total_kernel_stats()
                eng.set_option("overlap", 1)
                ser = {}
                for n in k1:
                    d = {f: k1[n][f] - k0.get(n, {}).get(f, 0.0) for f in k1[n]}
                    if d["launches"] > 0 and d["alg_bytes"] > 0:
                        ser[n] = {"achieved": d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6, "frac": d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS,
                                  "launches": d["launches"], "ms_per_launch": d["ms"] / d["launches"], "alg_bytes_per_launch": d["alg_bytes"] / d["launches"]}
                out["roofline"]["per_kernel_serialised"] = ser
                tb = sum(v["alg_bytes_per_launch"] * v["launches"] for v in ser.values())
                tm = sum(v["ms_per_launch"] * v["launches"] for v in ser.values())
                out["roofline"]["serialised_all_kernels_GBps"] = tb / max(tm, 1e-9) / 1e6
                out["roofline"]["serialised_share_of_bytes"] = {n: v["alg_bytes_per_launch"] * v["launches"] / max(tb, 1.0) for n, v in ser.items()}
            except Exception as e:  # noqa: BLE001
                out["roofline"]["per_kernel_serialised"] = {"error": repr(e)}
        if full is not None:
            out["full_stream"] = full
        if world > 1:
            lo = a.warmup * G
            cost = eng.estimate_costs(to_var[qv[lo:lo + G]][:, None], to_var[ev[lo:lo + G]])
            out["shard_cost_imbalance"] = {
                "count_split": sharding.imbalance(cost, [sharding.shard_range(G, world, r) for r in range(world)]),
                "cost_split": sharding.imbalance(cost, sharding.cost_balanced_ranges(cost, world)),
                "note": "max / mean shard cost (planner estimate) of one step's global batch, outside the clock; the C3 stream is "
                        "i.i.d., so equal counts already balance to ~1 % - and the estimate itself (0.1 s per 65 536 requests on "
                        "every rank) would cost more inside a step than it can win: --balance cost is an option, count the default"}
        if world == 1 and not a.no_configs:
            try:
                out["configs"] = other_configs(device, with_cpu=not a.no_cpu)
            except Exception as e:  # the headline line must not die with a side measurement
                out["configs"] = {"error": repr(e)}
            # SURVEY 8(d)'s n_evidence variants of the C3 stream on this engine (the final kernels, the timed region's options)
            for ne in (1, 8, 16):
                try:
                    out["configs"][f"C3_n_evidence_{ne}"] = c3_variant(eng, to_var, ne, calls=24, warmup_calls=16, batch=a.batch)
                except Exception as e:  # noqa: BLE001
                    out["configs"][f"C3_n_evidence_{ne}"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu:
            # GPU posteriors of the first 200 requests of the stream (what the reference legs work on)
            n_ref = 200
            post200 = eng.query_fixed(to_var[qv[:n_ref]][:, None], to_var[ev[:n_ref]], ec[:n_ref])
            ref = cpu_reference(a.cpu_seconds, a.cpu_procs) if a.cpu_seconds > 0 else None
            port, err_port = cpu_port(spec, qv[:n_ref], ev[:n_ref], ec[:n_ref], post200, a.port_seconds)
            if ref is not None and "error" not in ref[0]:
                single, aggregate, answers = ref
                err_ref, n_cmp = 0.0, 0
                for i, (index, values) in answers.items():
                    dense = np.zeros(4)
                    for key, v in zip(index, values):
                        dense[int(key[0])] = v
                    err_ref = max(err_ref, float(np.max(np.abs(dense - post200[i]))))
                    n_cmp += 1
                # the requests the live reference had to abandon at the cap: its own answers from the build container, where it
                # was given the ten minutes they take (tests/golden/c3_first16.json, written by tests/golden/make_c3_first16.py)
                n_gold = 0
                try:
                    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_first16.json")))["requests"]
                    for i, g in enumerate(gold[:a.cpu_procs]):
                        if i in answers:
                            continue
                        assert g["query"] == int(qv[i]) and [e for e, _ in g["evidence"]] == ev[i].tolist() and [c for _, c in g["evidence"]] == ec[i].tolist()
                        dense = np.zeros(4)
                        for key, h in zip(g["index"], g["values_hex"]):
                            dense[int(key[0])] = float.fromhex(h)
                        err_ref = max(err_ref, float(np.max(np.abs(dense - post200[i]))))
                        n_gold += 1
                except OSError:
                    pass
                out["cpu_baseline"] = single
                out["cpu_baseline"]["aggregate"] = aggregate
                out["max_abs_marginal_err_vs_reference"] = {"value": err_ref, "requests_compared": n_cmp + n_gold,
                                                            "against_the_live_reference": n_cmp,
                                                            "against_its_committed_answers_for_the_requests_abandoned_at_the_cap": n_gold}
                out["cpu_port"] = port
            else:  # oracle/_ref did not travel (fresh clone without `make -C oracle _ref`): the port is all there is
                out["cpu_baseline"] = port
                out["cpu_baseline"]["note"] = "oracle/_ref unavailable on this box: " + (ref[0]["error"] if ref else "not built")
            out["max_abs_marginal_err_vs_oracle"] = err_port
        if world == 1 and not a.no_configs and not a.threads:
            # a rank with ONE / TWO / FOUR planner threads (8 ranks sharing a small CPU quota): a fresh engine whose pool has that many
            # workers; the adaptive policy hands planning to order_kernel + emit_kernel where the host's workers would bound the
            # pipeline (the main engine's arena is released first).  Calls of 32 768: such a rank's shard of a 2^18-request step on
            # eight GPUs.  C3_two_planner_threads keeps its round-3 name.
            eng.close()
            for nt, name in ((2, "C3_two_planner_threads"), (1, "C3_planner_threads_1"), (4, "C3_planner_threads_4")):
                try:
                    bn2 = netspec.build(spec, sorobn_amd.BayesNet).use_device(device)
                    eng2 = bn2.backend.engine
                    eng2.set_option("threads", nt)
                    eng2.set_option("adaptive", 1)
                    for kv in a.opt:
                        k, v = kv.split("=")
                        eng2.set_option(k, float(v))
                    out["configs"][name] = c3_variant(eng2, to_var, a.n_evidence, calls=64, warmup_calls=40, batch=32768)  # (forty warm-up calls: the share controller and the pinned buffers of a fresh engine settle over the first twenty - r06_o / r06_q; sixty-four timed calls: the timed region starts with an idle GPU and the first call's planning is not hidden - a sixty-fourth of the region, the standalone runs' share)  # (a fresh engine: its buffers, the share controller)
                    eng2.close()
                except Exception as e:  # noqa: BLE001
                    out["configs"][name] = {"error": repr(e)}
            out["projected_8gpu"] = project_8gpu(out)
            # the drop-in pandas boundary and the section-8(f) rows last, on engines of their own (session A of round 5 ran them in front
            # of the n_evidence variants: their engines' device memory cut the main engine's arena budget and n_evidence = 8 ran in
            # several arena waves per call)
            try:
                bnp = netspec.build(spec, sorobn_amd.BayesNet).use_device(device)
                out["configs"]["C3_query_many_pandas"] = c3_pandas(bnp)
                bnp.backend.engine.close()
            except Exception as e:  # noqa: BLE001
                out["configs"]["C3_query_many_pandas"] = {"error": repr(e)}
            try:
                out["configs"].update(f_configs(device, with_cpu=not a.no_cpu))
            except Exception as e:  # noqa: BLE001
                out["configs"]["F_rows"] = {"error": repr(e)}
        out["summary"] = summary_of(out)  # LAST key: what a 2 000-byte tail of the line still shows
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()


def summary_of(out):
    """The figures a reader of the line's END needs (the driver records the last 2 000 bytes of a ~20 KB line):
    compact keys, rounded numbers, < 1 500 bytes.  q = queries/s, gb = gpu_bound (the VE kernels' busy time >= 0.9 x wall), gbp = the same counting the
    device planner's kernel, dev = device-planned requests per call."""
    cf = out.get("configs", {}) if isinstance(out.get("configs"), dict) else {}

    def g(name, key, nd=0):
        v = cf.get(name, {}).get(key) if isinstance(cf.get(name), dict) else None
        return None if v is None else (round(v, nd) if nd else (int(round(v)) if isinstance(v, float) else v))

    def variant(name):
        return {"q": g(name, "queries_per_s"), "gb": g(name, "gpu_bound"), "gbp": g(name, "gpu_bound_counting_the_planner"), "GBps": g(name, "all_kernels_GBps"),
                "dev": g(name, "device_planned_requests_per_call")}

    r = out["roofline"]
    ser = r.get("per_kernel_serialised", {}) if isinstance(r.get("per_kernel_serialised"), dict) else {}
    s = {"q": int(round(out["value"])), "ms": round(out["ms_per_step"], 1), "frac": round(r["frac"], 3), "all_GBps": int(round(r["all_kernels_GBps"])),
         "MB_q": round(r["alg_bytes_per_query"] / 1e6, 2), "gb": out["pipeline_clocks_ms_per_step"]["gpu_bound"],
         "ser_frac": {k.replace("ve_", "").replace("_kernel", ""): round(v["frac"], 3) for k, v in ser.items() if isinstance(v, dict) and "frac" in v},
         "C1_ms": g("C1_alarm_single_query", "ms_per_query", 3), "C2_q": g("C2_asia_100k", "queries_per_s"),
         "C5_ms": g("C5_gibbs_128_chains_x_100k", "wall_ms", 1), "C5_err": g("C5_gibbs_128_chains_x_100k", "max_abs_err_vs_exact", 5),
         "nev1": variant("C3_n_evidence_1"), "nev8": variant("C3_n_evidence_8"), "nev16": variant("C3_n_evidence_16"),
         "thr1": variant("C3_planner_threads_1"), "thr2": variant("C3_two_planner_threads"), "thr4": variant("C3_planner_threads_4"),
         "pandas_q": g("C3_query_many_pandas", "queries_per_s")}
    if "projected_8gpu" in out:
        s["proj8"] = round(out["projected_8gpu"]["scaling_vs_this_line"], 2)
    if "cpu_baseline" in out:
        s["ref_q_1core"] = round(out["cpu_baseline"].get("value", 0.0), 4)
    if "max_abs_marginal_err_vs_reference" in out:
        s["err_ref"] = out["max_abs_marginal_err_vs_reference"]["value"]
    if "shared_prefix" in r:
        s["MB_q_indep"] = round(r["shared_prefix"].get("independent_alg_bytes_per_query", 0.0) / 1e6, 2)
    return s


def run_c5(a, rank, world, local_rank, device, backend):
    """BASELINE config 5: 50-node K=8 grid, algorithm='gibbs', 100 k single-site updates per chain, 128 chains per GPU
    (1024 on 8 GPUs) of ONE chain stream, histograms summed onto rank 0 with mibn_comm_reduce_i64."""
    import netspec
    import sorobn_amd
    from sorobn_amd import sharding

    chains_per_gpu, iters = 128, 100_000
    spec5 = netspec.grid_spec(5, 10, 8, seed=0)
    bn5 = netspec.build(spec5, sorobn_amd.BayesNet).use_device(device)
    be = bn5.backend
    eng = be.engine
    comm, transport = make_comm(backend, world, rank, local_rank, eng)
    rng = np.random.default_rng(1)
    ev5 = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
    q, evs, codes = be.encode(("025",), ev5)
    free = [v for v in range(len(be.flat.names)) if v not in set(evs)]
    cycle = sorted(free, key=lambda v: be.flat.names[v])
    exact = bn5.query("025", event=ev5).to_numpy()
    total_chains = chains_per_gpu * world

    def step(s):
        return sharding.gibbs_sharded(eng, comm, q, evs, codes, total_chains, iters, seed=1000 + s, cycle=cycle)

    for s in range(a.warmup):
        step(s)
    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    hist = None
    for s in range(a.warmup, a.warmup + a.steps):
        h = step(s)
        hist = h if hist is None else hist + h
    comm.barrier()
    eng.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = float(comm.allreduce_max([dt])[0])
    if rank == 0:
        est = hist / float(hist.sum())
        out = {"metric": "Gibbs single-site updates/sec on 50-node 8-state grid BN (config 5)",
               "value": a.steps * total_chains * iters / dt, "unit": "updates/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "C5: 5x10 grid BN, 8 states/node, query node 25, 5 evidence nodes; "
                                      f"{iters} single-site updates x {chains_per_gpu} chains per GPU",
                          "parallelism": f"dp{world} (chain shards of one Philox stream, int64 histogram reduce)"},
               "roofline": {"bound": "latency/LDS (CPTs resident in LDS): HBM roofline n/a", "achieved": None, "peak": None,
                            "unit": None, "frac": None, "traffic": None},
               "max_abs_err_vs_exact": float(np.max(np.abs(est - exact))),
               # the pooled int64 histogram of the timed steps (seeds 1000 + step): any split of the chains over ranks reproduces the
               # unsharded one bit for bit (mibn_gibbs_shard: the Philox key is the global chain index) - the dry run checks exactly that
               "histogram": [int(x) for x in hist], "chains_total": total_chains, "seeds": [1000 + s for s in range(a.warmup, a.warmup + a.steps)],
               "reduce": {"none": "none", "rccl": "RCCL via the C-ABI (mibn_comm_reduce_i64), no PyTorch",
                          "files": "DRY RUN (MIBN_BENCH_BACKEND=files): the reduce through files"}[transport]}
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
