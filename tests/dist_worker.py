"""Worker of tests/test_dist.py: world_size-2 gloo run of the multi-GPU path's logic (sorobn_amd/sharding.py) - count-
and cost-balanced request shards + the posterior gather, chain shards + the int64 histogram reduce, max-over-ranks -
with the CPU plan simulator standing in for the GPU engine and torch.distributed/gloo (tests/torchcomm.py)
standing in for RCCL."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import netspec  # noqa: E402
import simengine  # noqa: E402
import sorobn_amd  # noqa: E402
from sorobn_amd import _capi, sharding  # noqa: E402
from torchcomm import TorchComm  # noqa: E402


class FakeGibbsEngine:
    """Deterministic per-chain histograms (a function of the global chain index, like the Philox-keyed kernel)."""

    def gibbs(self, qvars, evars, ecodes, n_chains, n_iterations, seed=0, cycle=None, chain_first=0):
        out = np.zeros(8, np.int64)
        for c in range(chain_first, chain_first + n_chains):
            out += np.random.default_rng([seed, c]).multinomial(n_iterations, np.full(8, 1 / 8))
        return out


def main():
    dist.init_process_group("gloo")
    comm = TorchComm()
    rank, world = comm.rank, comm.world
    spec = netspec.grid_spec(5, 5, 4, seed=0)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    be = bn.backend
    n = 61  # not divisible by the world size on purpose
    q, ev, ec = netspec.c3_requests(25, 4, n, 3, seed=5)
    to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(25)], np.int32)
    ref = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec) if rank == 0 else None

    # 1. count-balanced contiguous shards, gathered through a Comm
    lo, hi = sharding.shard_range(n, world, rank)
    local = be.engine.query_fixed(to_var[q[lo:hi]][:, None], to_var[ev[lo:hi]], ec[lo:hi])
    for full in (sharding.gather_posteriors(local, n, comm),):
        assert full.shape == (n, 4)
        if rank == 0:
            assert np.array_equal(full, ref)

    # 2. cost-balanced shards from the planner's estimates (mibn_estimate_costs on a planner-only context)
    f = be.flat
    planner = _capi.Engine(planner_only=True)
    planner.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.values)
    cost = planner.estimate_costs(to_var[q][:, None], to_var[ev])
    assert cost.shape == (n,) and (cost > 0).all()
    ranges = sharding.cost_balanced_ranges(cost, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert sharding.imbalance(cost, ranges) <= sharding.imbalance(cost, [sharding.shard_range(n, world, r) for r in range(world)]) + 1e-12
    lo, hi = ranges[rank]
    local = be.engine.query_fixed(to_var[q[lo:hi]][:, None], to_var[ev[lo:hi]], ec[lo:hi])
    full = sharding.gather_posteriors(local, n, comm, ranges=ranges)
    if rank == 0:
        assert np.array_equal(full, ref)

    # 2b. BASELINE config 4 as bench.py runs it (sharding.ShardedStream): the SAME requests whatever the world size, steps of a
    #     fixed global batch, contiguous shards (count- and cost-balanced), sub-batches per call, ONE gather per step; and the
    #     whole stream as one step with one gather (bench.py --full-stream)
    tq, tev = to_var[q][:, None], to_var[ev]
    for balance, G, sub in (("count", 20, 7), ("cost", 20, 64), ("count", n, 16)):
        be.engine.estimate_costs = planner.estimate_costs  # (the simulator has no planner of its own: same network, same ids)
        stream = sharding.ShardedStream(be.engine, comm, tq, tev, ec, G, sub_batch=sub, balance=balance)
        steps = range((n + G - 1) // G)
        res = stream.run(steps, keep=True)
        assert res["requests"] == n and abs(res["mass"] - n) < 1e-9 and int(stream.shard_requests.sum()) == n
        full = np.concatenate([res["gathered"][st] for st in steps], axis=0)
        if rank == 0:
            assert np.array_equal(full, ref), balance
        if G == n:  # one step: one gather; every rank holds every answer
            assert len(res["gathered"]) == 1 and res["first"].shape == (n, 4)

    # 3. Gibbs: chain shards of one stream, int64 histogram reduce onto rank 0 (bayes_net.py:736-737 on N GPUs)
    fake = FakeGibbsEngine()
    hist = sharding.gibbs_sharded(fake, comm, [0], [], [], 37, 1000, seed=4)
    if rank == 0:
        assert np.array_equal(hist, fake.gibbs([0], [], [], 37, 1000, seed=4)) and hist.sum() == 37 * 1000

    # 4. max over ranks (the bench's step time) and the barrier
    assert comm.allreduce_max([float(rank), 5.0 - rank]).tolist() == [float(world - 1), 5.0]
    comm.barrier()
    if rank == 0:
        with open(os.environ["DIST_OK_FILE"], "w") as fh:
            fh.write(f"ok {world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
