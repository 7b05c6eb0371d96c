"""TEST INFRASTRUCTURE - runs the *unmodified reference* (`BayesNet.query`, sorobn/bayes_net.py:796-875 ->
`_variable_elimination` 739-794) over a shard of a request stream and reports per-request wall times and answers.

Spawned by `bench.py`'s `cpu_baseline` leg (kind "reference") as independent single-threaded processes - the reference
cannot use more than one core - with PYTHONHASHSEED=0; `oracle.refload` supplies the module (from /root/reference, or from
the byte-compiled `oracle/_ref/` on the GPU box), the `vose` stub and the hash-ordered names that make the reference's
set-iteration elimination order (bayes_net.py:766, 779) row-major, without which it cannot finish a 5x5 grid.

    python oracle/ref_worker.py --workload c3 --first 200 --shard 0 --nshards 1 --budget 20

Requests [0, first) of the stream, those with index % nshards == shard, in stream order, until `budget` seconds of wall
time are spent (the request in flight at that moment is abandoned and reported as unfinished).  Output: one JSON line per
finished request {"i", "s", "values"} and a closing {"done": true, "elapsed", "finished", "attempted", "setup_s"}.
"""
import argparse
import json
import os
import signal
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class _Budget(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=["c3", "c1", "c2", "f1", "f2", "f3", "f4"])
    ap.add_argument("--rows", type=int, default=0, help="f1 / f3 / f4: rows of the data frame; f2: likelihood-weighting samples")
    ap.add_argument("--first", type=int, default=200)
    ap.add_argument("--shard", type=int, default=0)
    ap.add_argument("--nshards", type=int, default=1)
    ap.add_argument("--budget", type=float, default=20.0)
    ap.add_argument("--n-evidence", type=int, default=4)
    a = ap.parse_args()

    t_setup = time.perf_counter()
    import netspec
    from oracle import refload

    sorobn = refload.load()
    if a.workload in ("f1", "f2", "f3", "f4"):
        return section_8f(a, sorobn, netspec, time.perf_counter() - t_setup)
    if a.workload == "c3":
        spec = netspec.grid_spec(10, 10, 4, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        qv, ev, ec = netspec.c3_requests(100, 4, a.first, a.n_evidence, seed=1)
        name = lambda i: refload.HashedName(f"{int(i):03d}")
        reqs = [((name(qv[i]),), {name(ev[i, k]): int(ec[i, k]) for k in range(ev.shape[1])}) for i in range(a.first)]
    elif a.workload == "c2":
        bn = sorobn.examples.asia()
        reqs = [((q,), e) for q, e in netspec.asia_requests(list(bn.nodes), a.first, seed=0)]
    else:
        bn = sorobn.examples.alarm()
        reqs = [(("Burglary",), {"Mary calls": True, "John calls": True})] * a.first
    setup_s = time.perf_counter() - t_setup

    def on_alarm(signum, frame):
        raise _Budget()

    signal.signal(signal.SIGALRM, on_alarm)
    t0 = time.perf_counter()
    finished = attempted = 0
    signal.setitimer(signal.ITIMER_REAL, a.budget)
    try:
        for i in range(a.shard, a.first, a.nshards):
            attempted += 1
            t1 = time.perf_counter()
            ans = bn.query(*reqs[i][0], event=reqs[i][1])
            dt = time.perf_counter() - t1
            finished += 1
            idx = [list(k) if isinstance(k, tuple) else [k] for k in ans.index.tolist()]
            print(json.dumps({"i": i, "s": dt, "index": [[x if isinstance(x, (bool, int, str)) else int(x) for x in k] for k in idx],
                              "values": [float(v) for v in ans.to_numpy()]}), flush=True)
    except _Budget:
        pass
    signal.setitimer(signal.ITIMER_REAL, 0)
    print(json.dumps({"done": True, "elapsed": time.perf_counter() - t0, "finished": finished, "attempted": attempted,
                      "setup_s": setup_s, "reference": getattr(sorobn, "_mibn_refload", "?")}), flush=True)


def section_8f(a, sorobn, netspec, setup_s):
    """The SURVEY section 8(f) rows on the unmodified reference, one call each, on the inputs bench.py's `f_configs` gives the GPU
    path (same generators, same seeds): f1 `predict_proba` (bayes_net.py:934-962, through `full_joint_dist` 398-465) on a 3x3
    K=4 grid; f2 `query(algorithm="likelihood")` (621-663) on Asia; f3 `fit` (467-516) on the 10x10 K=4 grid's structure; f4
    `structure.chow_liu` (structure.py:9-63) on 100 four-state columns.  Output: one closing JSON line with `units` (rows or
    samples) and `elapsed`; a call that does not finish inside `--budget` reports `finished` 0."""
    import numpy as np
    import pandas as pd

    def frame(n_rows, names, K, seed):
        return pd.DataFrame(np.random.default_rng(seed).integers(0, K, (n_rows, len(names))), columns=names)

    if a.workload == "f1":
        spec = netspec.grid_spec(3, 3, 4, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet)
        X = frame(a.rows, list(spec["nodes"]), 4, 11)
        call = lambda: float(bn.predict_proba(X).sum())
    elif a.workload == "f2":
        bn = sorobn.examples.asia()
        call = lambda: bn.query("Lung cancer", event={"Smoker": True, "Dispnea": True}, algorithm="likelihood", n_iterations=a.rows).tolist()
    elif a.workload == "f3":
        spec = netspec.grid_spec(10, 10, 4, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet)
        X = frame(a.rows, [f"{i:03d}" for i in range(100)], 4, 12)
        call = lambda: len(bn.fit(X).P)
    else:
        X = frame(a.rows, [f"{i:03d}" for i in range(100)], 4, 13)
        call = lambda: len(sorobn.structure.chow_liu(X))

    def on_alarm(signum, frame_):
        raise _Budget()

    signal.signal(signal.SIGALRM, on_alarm)
    signal.setitimer(signal.ITIMER_REAL, a.budget)
    t0 = time.perf_counter()
    finished, result = 0, None
    try:
        result = call()
        finished = 1
    except _Budget:
        pass
    signal.setitimer(signal.ITIMER_REAL, 0)
    print(json.dumps({"done": True, "elapsed": time.perf_counter() - t0, "finished": finished, "attempted": 1, "units": a.rows, "result": result,
                      "setup_s": setup_s, "reference": getattr(sorobn, "_mibn_refload", "?")}), flush=True)


if __name__ == "__main__":
    main()
