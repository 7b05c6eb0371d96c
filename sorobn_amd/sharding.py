"""Multi-GPU sharding of independent requests.

Requests are independent given the (tiny, replicated) CPT tensors, so the request list is split into
contiguous shards, one process per GPU, with no data-path collective; the only communication is the
final all-gather of the dense posteriors (RCCL over xGMI when the process group's backend is "nccl";
the same code runs on "gloo" for the CPU tests).  torch.distributed is plumbing here, it never
touches the kernels.
"""
import numpy as np


def shard_range(n: int, world: int, rank: int):
    """Contiguous balanced shard [lo, hi) of n items for `rank` of `world`."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_posteriors(local: np.ndarray, n_total: int, group=None):
    """All-gather row-shards produced with `shard_range` back into the full [n_total, cells] array
    (every rank gets it).  Shards may differ by one row; they are padded to equal length for the
    collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    rows = -(-n_total // world)
    cells = local.shape[1] if local.ndim == 2 else 1
    buf = torch.zeros((rows, cells), dtype=torch.float64, device=dev)
    if len(local):
        buf[:len(local)] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(len(local), cells)).to(dev)
    out = torch.empty((world, rows, cells), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), buf.view(-1), group=group)
    out = out.cpu().numpy()
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        parts.append(out[r, :hi - lo])
    return np.concatenate(parts, axis=0)
