"""Worker of tests/test_dist.py: world_size-2 gloo run of the multi-GPU path's sharding + gather
(sorobn_amd/sharding.py), with the CPU plan simulator standing in for the GPU engine."""
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
from sorobn_amd.sharding import gather_posteriors, shard_range  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    spec = netspec.grid_spec(5, 5, 4, seed=0)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    be = bn.backend
    n = 61  # not divisible by the world size on purpose
    q, ev, ec = netspec.c3_requests(25, 4, n, 3, seed=5)
    to_var = np.array([be.flat.id[f"{i:03d}"] for i in range(25)], np.int32)
    lo, hi = shard_range(n, world, rank)
    local = be.engine.query_fixed(to_var[q[lo:hi]][:, None], to_var[ev[lo:hi]], ec[lo:hi])
    full = gather_posteriors(local, n)
    assert full.shape == (n, 4)
    if rank == 0:
        ref = be.engine.query_fixed(to_var[q][:, None], to_var[ev], ec)
        assert np.array_equal(full, ref)
        with open(os.environ["DIST_OK_FILE"], "w") as f:
            f.write(f"ok {world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
