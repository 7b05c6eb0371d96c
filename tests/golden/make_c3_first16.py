"""Golden answers of the UNMODIFIED reference for the FIXED request set of bench.py's `cpu_baseline` leg: requests 0..15 of the C3
stream (netspec.c3_requests(100, 4, 16, 4, seed=1)) on the 10x10 K=4 grid.  Build container only (the reference does not exist on
the GPU box):

    PYTHONHASHSEED=0 python tests/golden/make_c3_first16.py [--cap 3000] [--collect DIR]

Runs oracle/ref_worker.py once per request (one single-threaded process each, like the bench leg, but with a cap that lets every
request finish: the slowest take ~10 minutes in the reference's row-major elimination order) and writes tests/golden/c3_first16.json:
per request the index rows, the values as float.hex() and the reference's wall seconds here.  bench.py compares the GPU posteriors
with the live reference for the requests that finish under its 60 s cap and with THIS file for the ones it has to abandon, so that
`max_abs_marginal_err_vs_reference.requests_compared` is always 16.  --collect DIR assembles the file from worker outputs
(out_<i>.json) of an earlier run of the same commands instead of spawning them again."""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
N = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cap", type=float, default=3000.0)
    ap.add_argument("--collect", default="")
    a = ap.parse_args()
    outs = []
    if a.collect:
        for i in range(N):
            with open(os.path.join(a.collect, f"out_{i}.json")) as f:
                outs.append(f.read())
    else:
        assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        worker = os.path.join(ROOT, "oracle", "ref_worker.py")
        procs = [subprocess.Popen([sys.executable, worker, "--workload", "c3", "--first", str(N), "--shard", str(i), "--nshards", str(N),
                                   "--budget", str(a.cap)], env=env, stdout=subprocess.PIPE, text=True) for i in range(N)]
        outs = [p.communicate()[0] for p in procs]
    reqs = {}
    for i, so in enumerate(outs):
        for line in so.splitlines():
            d = json.loads(line)
            if "i" in d:
                assert d["i"] == i
                reqs[i] = {"index": d["index"], "values_hex": [float(v).hex() for v in d["values"]], "ref_seconds": round(d["s"], 2)}
    assert sorted(reqs) == list(range(N)), f"unfinished: {sorted(set(range(N)) - set(reqs))}"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import netspec
    q, ev, ec = netspec.c3_requests(100, 4, N, 4, seed=1)
    for i in range(N):
        reqs[i]["query"] = int(q[i])
        reqs[i]["evidence"] = [[int(e), int(c)] for e, c in zip(ev[i], ec[i])]
    with open(os.path.join(HERE, "c3_first16.json"), "w") as f:
        json.dump({"stream": "netspec.c3_requests(100, 4, 16, 4, seed=1) on netspec.grid_spec(10, 10, 4, seed=0)",
                   "requests": [reqs[i] for i in range(N)]}, f, indent=1)
    print("wrote c3_first16.json; reference seconds:", [reqs[i]["ref_seconds"] for i in range(N)])


if __name__ == "__main__":
    main()
