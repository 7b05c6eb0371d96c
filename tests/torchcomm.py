"""Test infrastructure: the `sharding` transport interface (`rank`, `world`, `allgather`, `reduce_i64`, `allreduce_max`, `barrier`,
`close`) on a torch.distributed process group - "gloo" on CPU - so that the world-size-2 tests of the N > 1 logic run in the GPU-less
build container.  Lives under tests/: the product package and bench.py never import PyTorch (the product transport is
`sharding.RcclComm` = mibn_comm_* of the C-ABI; the dry run is `sharding.FileComm`)."""
import numpy as np


class TorchComm:
    """Test hook: the same interface on a torch.distributed group (gloo on CPU, or nccl = RCCL through PyTorch)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def allgather(self, rows):
        torch = self._torch
        rows = np.ascontiguousarray(rows, np.float64)
        mine = torch.from_numpy(rows.reshape(-1)).to(self.dev)
        out = torch.empty(self.world * mine.numel(), dtype=torch.float64, device=self.dev)
        self._dist.all_gather_into_tensor(out, mine, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + rows.shape)

    def reduce_i64(self, arr, root=0):
        torch = self._torch
        t = torch.from_numpy(np.ascontiguousarray(arr, np.int64).copy()).to(self.dev)
        self._dist.reduce(t, dst=root, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy() if self.rank == root else np.ascontiguousarray(arr, np.int64).copy()

    def allreduce_max(self, values):
        torch = self._torch
        t = torch.from_numpy(np.ascontiguousarray(values, np.float64).copy().reshape(-1)).to(self.dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def barrier(self):
        self._dist.barrier(group=self.group)

    def close(self):
        pass
