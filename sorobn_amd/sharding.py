"""Multi-GPU sharding of independent requests and Gibbs chains.

Requests are independent given the (tiny, replicated) CPT tensors and Gibbs chains are independent given a seed, so
the work is split into contiguous shards, one process per GPU, with NO data-path collective.  The only communication is
the final gather of the dense posteriors (exact path: one all-gather of [shard x K^n_query] float64) or the sum of the
int64 histograms (Gibbs: one reduce of [K^n_query], replacing the value_counts / normalise of bayes_net.py:736-737).

Shards are balanced by *cost*, not by count, where costs differ (`cost_balanced_ranges` over the planner's section-8(d)
byte estimates, `Engine.estimate_costs`): one request of the 10x10 grid costs between 10 KB and 150 MB.

Transports behind one small interface (`rank`, `world`, `allgather`, `reduce_i64`, `allreduce_max`, `barrier`):
  * `RcclComm`  - the product path: the C-ABI's mibn_comm_* entry points, directly on RCCL over xGMI, no PyTorch.  The
                  128-byte RCCL id travels from rank 0 to the other ranks of the node through a file.
  * `TorchComm` - a test hook: the same calls on a torch.distributed process group ("gloo" on CPU: the world-size-2
                  tests of the N > 1 logic in the GPU-less build container).
  * `SoloComm`  - world size 1.
"""
import os
import tempfile
import time

import numpy as np


# ------------------------------------------------------------------------------------------------ splitting

def shard_range(n: int, world: int, rank: int):
    """Contiguous count-balanced shard [lo, hi) of n items for `rank` of `world`."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def cost_balanced_ranges(costs, world: int):
    """Contiguous shards [(lo, hi)] * world of len(costs) items whose cost sums are as equal as a contiguous split
    allows: boundary r is the first prefix whose cost reaches r / world of the total (ties and zero totals fall back
    to the count split).  Deterministic, so every rank computes the same boundaries from the same estimates."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    total = float(costs.sum()) if n else 0.0
    if world <= 1 or n == 0 or not np.isfinite(total) or total <= 0.0:
        return [shard_range(n, world, r) for r in range(world)]
    prefix = np.cumsum(costs)
    targets = total * np.arange(1, world, dtype=np.float64) / world
    cuts = np.searchsorted(prefix, targets, side="left") + 1  # item that crosses the target stays left of the cut ...
    # ... unless stopping before it is closer to the target
    before = np.where(cuts - 2 >= 0, prefix[np.maximum(cuts - 2, 0)], 0.0)
    after = prefix[np.minimum(cuts - 1, n - 1)]
    cuts = np.where(np.abs(before - targets) <= np.abs(after - targets), cuts - 1, cuts)
    cuts = np.clip(np.maximum.accumulate(cuts), 0, n)
    bounds = [0, *cuts.tolist(), n]
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]


def imbalance(costs, ranges):
    """max shard cost / mean shard cost (1.0 = perfectly balanced)."""
    costs = np.asarray(costs, dtype=np.float64)
    sums = np.array([costs[lo:hi].sum() for lo, hi in ranges])
    return float(sums.max() / sums.mean()) if sums.mean() > 0 else 1.0


# ------------------------------------------------------------------------------------------------ transports

class SoloComm:
    rank, world = 0, 1

    def allgather(self, rows):
        return np.asarray(rows, np.float64)[None]

    def reduce_i64(self, arr, root=0):
        return np.asarray(arr, np.int64).copy()

    def allreduce_max(self, values):
        return np.asarray(values, np.float64).copy()

    def barrier(self):
        pass

    def close(self):
        pass


def _launcher_start():
    """Epoch seconds at which this rank's parent process (the launcher: bench.py --gpus N, torchrun's agent, a shell) started -
    from /proc, so that every rank of a launch computes the same value; None where /proc is not readable."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])  # field 22: starttime, in clock ticks since boot
        with open("/proc/stat") as f:
            btime = next(float(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration, IndexError):
        return None


def _comm_dir():
    return os.environ.get("MIBN_COMM_DIR") or tempfile.gettempdir()


def _launch_tag():
    """What the ranks of ONE launch on a node share and no other launch does: the launcher's nonce (bench.py --gpus N creates
    a private directory and a random MIBN_LAUNCH_NONCE per launch), else the rendezvous port, the elastic run id and the
    launcher's pid."""
    nonce = os.environ.get("MIBN_LAUNCH_NONCE")
    if nonce:
        return nonce
    return "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                     os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.getppid()))


def _id_file():
    return os.path.join(_comm_dir(), f"mibn_comm_{_launch_tag()}.id")


def _fresh(path, t0):
    """A file of THIS launch: written after the launcher started (a crashed earlier launch with the same tag - same port, same
    recycled pid - cannot have done that); without /proc: not older than ten minutes."""
    start = _launcher_start()
    return os.path.getmtime(path) >= (start - 1.0 if start is not None else t0 - 600.0)


def exchange_id(rank, world, make_id, path=None, timeout_s=120.0):
    """Rank 0 calls make_id() and publishes the bytes (atomic rename, after removing whatever an earlier launch left under the
    same name); the others wait for a file of this launch."""
    path = path or _id_file()
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            if _fresh(path, t0):
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) >= 128:
                    return uid[:128], path
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout_s:.0f} s (is rank 0 alive?)")
        time.sleep(0.01)


def all_agree(rank, world, ok, what, timeout_s=120.0):
    """A collective boolean AND over the ranks of a node without a communicator (it decides whether one can be built): every
    rank leaves a marker file of this launch, then reads everybody's.  Returns True iff every rank reported ok; a rank that
    does not report within the timeout counts as a failure."""
    base = os.path.join(_comm_dir(), f"mibn_vote_{_launch_tag()}_{what}")
    mine = f"{base}.{rank}"
    tmp = f"{mine}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        f.write("1" if ok else "0")
    os.replace(tmp, mine)
    t0 = time.time()
    verdict = True
    for r in range(world):
        p = f"{base}.{r}"
        while True:
            try:
                if _fresh(p, t0):
                    with open(p) as f:
                        verdict = verdict and f.read().strip() == "1"
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout_s:
                verdict = False
                break
            time.sleep(0.01)
    return verdict


class RcclComm:
    """mibn_comm_* of include/mibn.h on the engine's own device and stream (RCCL over xGMI)."""

    def __init__(self, engine, rank=None, world=None):
        self.engine = engine
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        # What can fail on one rank only - loading librccl.so - happens BEFORE anybody enters the collective
        # ncclCommInitRank, and the ranks agree on the outcome: either all of them build the communicator or all of them
        # raise (and the caller falls back as a whole); a rank that failed alone would leave the others hanging in the init.
        try:
            probe = engine.comm_unique_id()  # (dlopen + ncclGetUniqueId; only rank 0's id is used)
            err = None
        except Exception as e:  # noqa: BLE001
            probe, err = None, e
        if not all_agree(self.rank, self.world, err is None, "rccl_load"):
            raise RuntimeError(f"mibn_comm_* unavailable on at least one rank of the node (this rank: {err!r})")
        uid, path = exchange_id(self.rank, self.world, lambda: probe)
        engine.comm_init(self.rank, self.world, uid)
        engine.comm_barrier()  # every rank has read the id
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    def allgather(self, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        return self.engine.comm_allgather(rows.reshape(-1), self.world).reshape((self.world,) + rows.shape)

    def reduce_i64(self, arr, root=0):
        return self.engine.comm_reduce_i64(arr, root)

    def allreduce_max(self, values):
        return self.engine.comm_allreduce_max(values)

    def barrier(self):
        self.engine.comm_barrier()

    def close(self):
        self.engine.comm_destroy()


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


# ------------------------------------------------------------------------------------------------ the two gathers

def gather_posteriors(local: np.ndarray, n_total: int, comm=None, ranges=None, group=None):
    """All-gather row shards back into the full [n_total, cells] array (every rank gets it).  `ranges` = the shard of
    every rank (default: the count split of `shard_range`); shards of different length are padded for the collective."""
    if comm is None:
        comm = TorchComm(group)  # (the pre-round-2 signature: a torch.distributed group)
    world = comm.world
    ranges = ranges or [shard_range(n_total, world, r) for r in range(world)]
    rows = max(hi - lo for lo, hi in ranges)
    local = np.asarray(local, np.float64)
    cells = local.shape[1] if local.ndim == 2 else 1
    buf = np.zeros((rows, cells), np.float64)
    if len(local):
        buf[:len(local)] = local.reshape(len(local), cells)
    out = comm.allgather(buf)
    return np.concatenate([out[r, :hi - lo] for r, (lo, hi) in enumerate(ranges)], axis=0)


def gibbs_sharded(engine, comm, qvars, evars, ecodes, n_chains, n_iterations, seed=0, cycle=None, root=0):
    """Config 5 on N GPUs: rank r runs chains shard_range(n_chains, world, r) of ONE stream (the Philox key is the global
    chain index, mibn_gibbs_shard) and the int64 histograms are summed onto `root` - bit for bit the histogram of the
    single-GPU call with n_chains chains.  Returns the histogram (meaningful on `root`)."""
    lo, hi = shard_range(int(n_chains), comm.world, comm.rank)
    counts = engine.gibbs(qvars, evars, ecodes, hi - lo, n_iterations, seed=seed, cycle=cycle, chain_first=lo)
    return comm.reduce_i64(counts, root=root)
