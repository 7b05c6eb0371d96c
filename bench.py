#!/usr/bin/env python
"""Benchmark of the exact-inference hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic requests: `--batch` exact posterior
queries (1 query node + 4 evidence nodes, the BASELINE C3 stream from default_rng(1)) on the
synthetic 10x10 grid BN with 4 states per node (Dirichlet(1) CPTs from default_rng(0)).  Inputs
(the flattened network) are resident in HBM before the timed region; the timed region covers
planning, program upload, the VE kernel, result download and - for N > 1 - the RCCL all-gather of
the posteriors.  Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), requests
are independent so every rank processes its own contiguous shard of the stream (weak scaling, no
data-path collective besides the final gather).

Rank 0 prints ONE JSON line with the driver's contract fields plus `roofline` (dominant kernel:
algorithmic bytes per launch / HIP-event duration, vs the 8 TB/s HBM peak) and `cpu_baseline` (the C
oracle = port of the reference's sparse VE, timed on a bounded sample of the same stream).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per launch come from the rocprofv3 PMC passes of this same command (separate --pmc FETCH_SIZE / WRITE_SIZE
# runs, corrected as MI355X_MICROARCH.md prescribes): they cannot be collected inside this process, so the bench
# reports the figure of the newest committed pass (profiles/r*_pmc.json) together with its provenance, or null.


def pmc_traffic(kernel):
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel") == kernel:
            best = (f, d)
    if not best:
        return None, None
    f, d = best
    return d["traffic_bytes_per_launch"], {"file": os.path.relpath(f, ROOT), "alg_bytes_per_launch_same_run": d.get("alg_bytes_per_launch_same_run")}


def cpu_baseline(spec, qv, ev, ec, gpu_post, budget_s, max_rows=3e7):
    """Time the C oracle (oracle/ve_oracle.c, kind="port": sparse-table VE restating
    bayes_net.py:739-794, eliminating in ascending-name = row-major order like the hash-ordered
    reference) on the first requests of the stream until `budget_s` seconds are spent; also returns the
    max-abs marginal error of the GPU posteriors on that sample.  Requests whose row-major product
    would exceed `max_rows` rows (minutes each on the CPU; the reference needs 448 s for the worst
    one) are skipped and counted - so the CPU figure is optimistic."""
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
            "sample": f"{n} of the first {n + skipped} requests of the C3 stream (rng seed 1) in {dt:.1f} s; the other "
                      f"{skipped} (> {max_rows:.0e} row-major product rows, minutes each on a CPU core) were not run, so "
                      "the figure is an upper bound for the whole stream; oracle/ve_oracle.c, single thread, "
                      "row-major (ascending name) elimination order like the hash-ordered reference"}, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32768, help="requests per step per GPU")
    ap.add_argument("--n-evidence", type=int, default=4)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sync", action="store_true", help="one blocking mibn_query_batch per step instead of the two-deep pipeline")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (experiments), e.g. --opt chunk=32768")
    ap.add_argument("--threads", type=int, default=0, help="planner threads of this rank (0 = host threads / ranks on the node)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        a.gpus = world

    import torch  # plumbing only: barrier / synchronize / RCCL gather
    import torch.distributed as dist

    # MIBN_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 path run on a box with fewer GPUs than ranks (ranks then
    # share devices and the collectives run on host tensors); the measured configuration is always nccl (= RCCL)
    backend = os.environ.get("MIBN_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    coll_dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import netspec
    import sorobn_amd

    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn_amd.BayesNet).use_device(device)
    be = bn.backend  # flatten + upload: network resident in HBM from here on
    eng = be.engine
    if a.threads:
        eng.set_option("threads", a.threads)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, float(v))

    total_steps = a.warmup + a.steps
    n_req = total_steps * world * a.batch
    qv, ev, ec = netspec.c3_requests(100, 4, n_req, a.n_evidence, seed=1)
    # names "000".."099" sort like the ids, but variable ids follow bn.nodes: map stream ids -> var ids
    to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(100)], np.int32)

    def shard(step):
        lo = (step * world + rank) * a.batch
        return to_var[qv[lo:lo + a.batch]], to_var[ev[lo:lo + a.batch]], ec[lo:lo + a.batch], lo

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = None
    if world > 1:
        gathered = torch.empty((world, a.batch, 4), dtype=torch.float64, device=coll_dev)

    # The K timed steps are pipelined two deep (mibn_submit_batch / mibn_wait): the host plans step s+1 while the
    # GPU runs step s, as a server streaming batches would.  Every step is complete - posteriors on the host and,
    # for N > 1, gathered over RCCL - before the closing barrier.
    def submit(step):
        q, e, c, lo = shard(step)
        return eng.submit_fixed(q[:, None], e, c), lo

    def finish(pending):
        handle, lo = pending
        post = eng.wait(handle)
        if world > 1:  # final gather of the posteriors over xGMI (RCCL)
            mine = torch.from_numpy(post).to(coll_dev, non_blocking=False)
            dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1))
        return post, lo

    def run(steps):
        first, pending = None, None
        for s in steps:
            nxt = submit(s)
            if a.sync:
                done = finish(nxt)
                first = first or done
                continue
            if pending is not None:
                done = finish(pending)
                first = first or done
            pending = nxt
        if pending is not None:
            done = finish(pending)
            first = first or done
        eng.drain()  # every launch finished and its HIP-event time booked
        return first

    run(range(a.warmup))
    barrier()
    st0, ks0 = eng.total_stats(), eng.total_kernel_stats()
    t0 = time.perf_counter()
    first_post, first_lo = run(range(a.warmup, total_steps))
    barrier()
    dt = time.perf_counter() - t0
    st1, ks1 = eng.total_stats(), eng.total_kernel_stats()
    agg = {k: st1[k] - st0[k] for k in st1}
    kagg = {n: {f: ks1[n][f] - ks0.get(n, {}).get(f, 0.0) for f in ks1[n]} for n in ks1}
    kagg = {n: d for n, d in kagg.items() if d["launches"] > 0}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        n_queries = a.steps * world * a.batch
        # dominant kernel = the specialisation with the most HIP-event time over the timed region
        dom = max(kagg, key=lambda n: kagg[n]["ms"])
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: 10x10 grid BN, 4 states/node, Dirichlet(1) CPTs rng(0); requests = "
                                   f"1 query + {a.n_evidence} evidence nodes, rng(1) stream",
                       "requests_per_step_per_gpu": a.batch, "parallelism": f"dp{world} (independent shards)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": dom, "alg_bytes_per_launch": bytes_per_launch,
                         "ms_per_launch": ms_per_launch, "launches": launches,
                         "share_of_kernel_time": kagg[dom]["ms"] / agg["kernel_ms"],
                         "all_kernels_GBps": all_kernels,
                         "alg_bytes_per_query": agg["alg_bytes"] / (a.steps * a.batch)},
            "kernels": {n: {"launches": d["launches"], "ms": d["ms"], "alg_GB": d["alg_bytes"] / 1e9,
                            "GBps": d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6}
                        for n, d in sorted(kagg.items(), key=lambda kv: -kv[1]["ms"])},
            "breakdown_ms_per_step": {k: agg[k] / a.steps for k in ("plan_ms", "h2d_ms", "kernel_ms", "d2h_ms", "total_ms")},
        }
        if world == 1 and not a.no_cpu:
            lo = first_lo
            cb, err = cpu_baseline(spec, qv[lo:lo + a.batch], ev[lo:lo + a.batch], ec[lo:lo + a.batch],
                                   first_post, a.cpu_seconds)
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(dom)
            out["cpu_baseline"] = cb
            out["max_abs_marginal_err_vs_oracle"] = err
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
