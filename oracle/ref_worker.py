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
    ap.add_argument("--workload", default="c3", choices=["c3", "c1", "c2"])
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


if __name__ == "__main__":
    main()
