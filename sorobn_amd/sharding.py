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
                  (The world-size-2 CPU tests run the same calls on a torch.distributed gloo group through `tests/torchcomm.py` -
                  test infrastructure: nothing in this package or in bench.py imports PyTorch.)
  * `FileComm`  - the dry run (`MIBN_BENCH_BACKEND=files`): N ranks on FEWER GPUs than ranks (RCCL refuses two ranks on one
                  device) run everything of the N > 1 path - the launch, the librccl probe on every rank, the vote, rank 0's
                  ncclGetUniqueId and the id exchange, the shards, the gather's packing - except ncclCommInitRank and the
                  collectives themselves, which go through files of the launch's private directory.  No PyTorch.
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


_ID_ERR = b"MIBN-ID-ERROR:"  # what rank 0 publishes in place of an id it could not create


def exchange_id(rank, world, make_id, path=None, timeout_s=120.0):
    """Rank 0 calls make_id() and publishes the bytes (atomic rename, after removing whatever an earlier launch left under the
    same name); the others wait for a file of this launch.  If make_id() raises, rank 0 publishes an error marker instead and the
    readers fail at once with its text."""
    path = path or _id_file()
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        try:
            uid = make_id()
        except Exception as e:  # the readers must not poll for two minutes for an id that will never come
            with open(tmp, "wb") as f:
                f.write(_ID_ERR + repr(e).encode()[:200])
            os.replace(tmp, path)
            raise
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
                if uid.startswith(_ID_ERR):
                    raise RuntimeError(f"rank {rank}: rank 0 could not create the RCCL id: {uid[len(_ID_ERR):].decode(errors='replace')}")
                if len(uid) >= 128:
                    return uid[:128], path
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout_s:.0f} s (is rank 0 alive?)")
        time.sleep(0.01)


_attempt = {}  # per process: how many times a vote / an id exchange of a given name has run (every rank constructs its
                # communicators in the same order, so the counters agree across the ranks of a launch)


def _next_attempt(what):
    _attempt[what] = _attempt.get(what, 0) + 1
    return _attempt[what]


def all_agree(rank, world, ok, what, timeout_s=120.0):
    """A collective boolean AND over the ranks of a node without a communicator (it decides whether one can be built): every
    rank leaves a marker file of this launch AND of this attempt (a second communicator built in the same launch must not read
    the first one's votes - ADVICE r3), then reads everybody's.  Returns (verdict, my file): True iff every rank reported ok; a
    rank that does not report within the timeout counts as a failure.  The caller unlinks its own file once every rank has read
    it (after the barrier that follows the init), see `RcclComm`."""
    base = os.path.join(_comm_dir(), f"mibn_vote_{_launch_tag()}_{what}_a{_next_attempt('vote_' + what)}")
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
    return verdict, mine


def _unlink(path):
    try:
        os.unlink(path)
    except OSError:
        pass


def _probe_and_vote(engine, rank, world):
    """Every rank checks that librccl.so loads (mibn_comm_probe: dlopen + symbols - no id, no listener thread, no socket on the
    ranks that are not rank 0), the ranks agree on the outcome, and every rank echoes its device and links once.
    -> this rank's vote file (to unlink after the first barrier)."""
    try:
        engine.comm_probe()
        err = None
    except Exception as e:  # noqa: BLE001
        err = e
    try:
        info = engine.device_info()
    except Exception as e:  # noqa: BLE001
        info = f"device info unavailable ({e!r})"
    import sys
    print(f"[mibn comm] rank {rank}/{world} pid {os.getpid()} {info}", file=sys.stderr, flush=True)
    ok, vote = all_agree(rank, world, err is None, "rccl_load")
    if not ok:  # (the vote file stays: a slower peer may still be reading it - the launcher's directory is cleaned anyway)
        raise RuntimeError(f"mibn_comm_* unavailable on at least one rank of the node (this rank: {err!r})")
    return vote


class RcclComm:
    """mibn_comm_* of include/mibn.h on the engine's own device and stream (RCCL over xGMI)."""

    def __init__(self, engine, rank=None, world=None):
        self.engine = engine
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        # What can fail on one rank only - loading librccl.so - happens BEFORE anybody enters the collective
        # ncclCommInitRank, and the ranks agree on the outcome: either all of them build the communicator or all of them
        # raise (there is no fallback transport: the launch fails as a whole); a rank that failed alone would leave the others hanging in the init
        # (which is bounded besides: mibn_comm_init gives up after MIBN_COMM_INIT_TIMEOUT_S seconds).
        vote = _probe_and_vote(engine, self.rank, self.world)
        uid, path = exchange_id(self.rank, self.world, engine.comm_unique_id,  # (ncclGetUniqueId on rank 0 only)
                                path=f"{_id_file()}.a{_next_attempt('id')}")
        # (on a failure the vote and id files stay where they are: a peer may still be polling for them, and a rank that raised
        #  here is on its way out - bench.py exits non-zero, there is no second transport behind this one)
        engine.comm_init(self.rank, self.world, uid)
        engine.comm_barrier()  # every rank has read the id and everybody's vote
        _unlink(vote)
        if self.rank == 0:
            _unlink(path)
        self.rccl_ranks, self.rccl_rank = engine.comm_count()  # what RCCL itself reports (ncclCommCount / ncclCommUserRank)
        if (self.rccl_ranks, self.rccl_rank) != (self.world, self.rank):
            raise RuntimeError(f"RCCL reports rank {self.rccl_rank} of {self.rccl_ranks}, the launch says {self.rank} of {self.world}")

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


class FileComm:
    """The dry run of the N > 1 path on a box with fewer GPUs than ranks (`MIBN_BENCH_BACKEND=files`; tests/test_dist.py): the
    same launch, probe, vote and id exchange as `RcclComm` - rank 0 really calls ncclGetUniqueId - but no ncclCommInitRank (RCCL
    refuses two ranks on one device) and collectives through files of the launch's private directory (MIBN_COMM_DIR).  Slow and
    single-node by construction: a check of the plumbing, never a measurement."""

    def __init__(self, engine, rank=None, world=None, timeout_s=300.0):
        self.engine = engine
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.timeout_s = timeout_s
        self._seq = 0
        self._mine = []
        self._base = os.path.join(_comm_dir(), f"mibn_filecomm_{_launch_tag()}_a{_next_attempt('filecomm')}")
        if engine is not None and not getattr(engine, "planner_only", False):
            vote = _probe_and_vote(engine, self.rank, self.world)
            uid, path = exchange_id(self.rank, self.world, engine.comm_unique_id, path=f"{_id_file()}.a{_next_attempt('id')}")
            assert len(uid) == 128
            self.barrier()
            _unlink(vote)
            if self.rank == 0:
                _unlink(path)

    def _exchange(self, arr):
        """Every rank publishes `arr` for this sequence number and reads everybody's.  -> list of arrays by rank."""
        self._seq += 1
        mine = f"{self._base}.{self._seq}.{self.rank}.npy"
        tmp = f"{mine}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            np.save(f, np.ascontiguousarray(arr))
        os.replace(tmp, mine)
        self._mine.append(mine)
        out, t0 = [], time.time()
        for r in range(self.world):
            p = f"{self._base}.{self._seq}.{r}.npy"
            while True:
                try:
                    out.append(np.load(p))
                    break
                except (OSError, ValueError, EOFError):
                    if time.time() - t0 > self.timeout_s:
                        raise TimeoutError(f"rank {self.rank}: rank {r} did not reach collective {self._seq} within {self.timeout_s:.0f} s")
                    time.sleep(0.002)
        # a file is read by every rank before any rank can publish sequence number + 2 (it has to pass + 1 first): drop old ones
        while len(self._mine) > 2:
            _unlink(self._mine.pop(0))
        return out

    def allgather(self, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        return np.stack(self._exchange(rows))

    def reduce_i64(self, arr, root=0):
        parts = self._exchange(np.ascontiguousarray(arr, np.int64))
        return np.sum(parts, axis=0).astype(np.int64) if self.rank == root else np.ascontiguousarray(arr, np.int64).copy()

    def allreduce_max(self, values):
        return np.max(np.stack(self._exchange(np.ascontiguousarray(values, np.float64).reshape(-1))), axis=0)

    def barrier(self):
        self._exchange(np.zeros(1))

    def close(self):
        # two barriers: whoever completes the second knows that every rank has read all files of the first (and older ones), which
        # can go; a rank's file of the LAST barrier may still be being read - it stays for the launcher's clean-up of the directory
        self.barrier()
        self.barrier()
        for p in self._mine[:-1]:
            _unlink(p)
        self._mine = []


# ------------------------------------------------------------------------------------------------ the two gathers

def gather_posteriors(local: np.ndarray, n_total: int, comm, ranges=None):
    """All-gather row shards back into the full [n_total, cells] array (every rank gets it).  `ranges` = the shard of
    every rank (default: the count split of `shard_range`); shards of different length are padded for the collective."""
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


class ShardedStream:
    """The exact path over a request stream on N ranks (BASELINE configs 3 and 4): the stream is cut into STEPS of `global_batch`
    requests; a step's requests are split into contiguous shards, one per rank (equal counts, or equal planner cost estimates
    with balance="cost"); a rank works through its shard in sub-batches of `sub_batch` requests, two calls in flight
    (mibn_submit_batch / mibn_wait: the host plans sub-batch k + 1 while the GPU runs k - across step boundaries too); a step
    ends with ONE all-gather of the dense posteriors (`gather_posteriors`), after which every rank holds the step's
    [global_batch x cells] answers.  No other communication.  `global_batch` fixed while N grows = strong scaling (config 4: the
    same requests on 2 / 4 / 8 GPUs); `global_batch` = N x per-rank batch = weak scaling.

    engine: `_capi.Engine` (or anything with query_fixed; submit_fixed / wait are used when present).  qvars [n, nq], evars
    [n, ne], ecodes [n, ne]: the stream, variable ids of the engine's network."""

    def __init__(self, engine, comm, qvars, evars, ecodes, global_batch, sub_batch=32768, balance="count", pipelined=True):
        self.engine, self.comm = engine, comm
        self.q, self.ev, self.ec = qvars, evars, ecodes
        self.G, self.sub = int(global_batch), max(1, int(sub_batch))
        self.balance = balance
        self.pipelined = pipelined and hasattr(engine, "submit_fixed")
        self.shard_requests = np.zeros(comm.world, np.int64)  # requests every rank has processed so far
        self.cells = None

    def ranges(self, step):
        """The shard of every rank inside step `step` (same on every rank: the estimates are deterministic)."""
        world = self.comm.world
        lo = step * self.G
        n = min(self.G, len(self.q) - lo)
        if self.balance == "cost" and world > 1:
            cost = self.engine.estimate_costs(self.q[lo:lo + n], self.ev[lo:lo + n])
            return cost_balanced_ranges(cost, world)
        return [shard_range(n, world, r) for r in range(world)]

    def run(self, steps, keep=False):
        """-> {"first": gathered posteriors of the first step run, "first_step", "requests", "mass" (sum of all gathered posteriors: =
        requests when every answer is a distribution), "gathered" (per step, only if keep)}."""
        comm, eng = self.comm, self.engine
        jobs = []  # (step, lo, hi, is_last_of_step, ranges)
        for st in steps:
            rg = self.ranges(st)
            lo, hi = st * self.G + rg[comm.rank][0], st * self.G + rg[comm.rank][1]
            cuts = list(range(lo, hi, self.sub)) or [lo]
            for k, a in enumerate(cuts):
                jobs.append((st, a, min(hi, a + self.sub), k == len(cuts) - 1, rg))
            self.shard_requests += np.array([b - a for a, b in rg], np.int64)
        out = {"first": None, "first_step": None, "requests": 0, "mass": 0.0, "gathered": {} if keep else None}
        parts = []

        def finish(job, handle):
            st, a, b, last, rg = job
            if b > a:
                post = eng.wait(handle) if self.pipelined else handle
                parts.append(np.asarray(post, np.float64).reshape(b - a, -1))
            if not last:
                return
            n = min(self.G, len(self.q) - st * self.G)
            if self.cells is None:  # cells per posterior: agreed once (a rank whose shard is empty has no answer to read it from)
                mine = parts[0].shape[1] if parts else 0
                self.cells = int(comm.allreduce_max([float(mine)])[0]) if comm.world > 1 else mine
            local = np.concatenate(parts, axis=0) if parts else np.zeros((0, self.cells))
            parts.clear()
            full = gather_posteriors(local, n, comm, ranges=rg) if comm.world > 1 else local
            out["requests"] += n
            out["mass"] += float(full.sum())
            if out["first"] is None:
                out["first"], out["first_step"] = full, st
            if keep:
                out["gathered"][st] = full

        pending = None
        for job in jobs:
            st, a, b, last, rg = job
            if self.pipelined:
                nxt = eng.submit_fixed(self.q[a:b], self.ev[a:b], self.ec[a:b]) if b > a else None  # (an empty shard: gather only)
                if pending is not None:
                    finish(*pending)
                pending = (job, nxt)
            else:
                finish(job, eng.query_fixed(self.q[a:b], self.ev[a:b], self.ec[a:b]) if b > a else None)
        if pending is not None:
            finish(*pending)
        return out


def gibbs_sharded(engine, comm, qvars, evars, ecodes, n_chains, n_iterations, seed=0, cycle=None, root=0):
    """Config 5 on N GPUs: rank r runs chains shard_range(n_chains, world, r) of ONE stream (the Philox key is the global
    chain index, mibn_gibbs_shard) and the int64 histograms are summed onto `root` - bit for bit the histogram of the
    single-GPU call with n_chains chains.  Returns the histogram (meaningful on `root`)."""
    lo, hi = shard_range(int(n_chains), comm.world, comm.rank)
    counts = engine.gibbs(qvars, evars, ecodes, hi - lo, n_iterations, seed=seed, cycle=cycle, chain_first=lo)
    return comm.reduce_i64(counts, root=root)
