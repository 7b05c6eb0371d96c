"""The N>1 path (contiguous shards, no data-path collective, one final all-gather) on CPU:
world_size 2, gloo backend."""
import os
import subprocess
import sys

import numpy as np

from sorobn_amd.sharding import cost_balanced_ranges, exchange_id, imbalance, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_cost_balanced_ranges():
    rng = np.random.default_rng(0)
    for n, world in ((0, 4), (1, 4), (3, 8), (1000, 2), (1000, 8), (32768, 8)):
        cost = rng.lognormal(0.0, 2.0, n)  # request costs vary ~1000x
        rg = cost_balanced_ranges(cost, world)
        assert len(rg) == world and rg[0][0] == 0 and rg[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rg, rg[1:])) and all(lo <= hi for lo, hi in rg)
        if n >= 1000:
            count = [shard_range(n, world, r) for r in range(world)]
            assert imbalance(cost, rg) <= imbalance(cost, count) + 1e-12
            assert imbalance(cost, rg) < 1.0 + 4 * cost.max() / (cost.sum() / world)
    # one dominant request: it gets a shard of its own
    cost = np.ones(100)
    cost[40] = 1000.0
    rg = cost_balanced_ranges(cost, 4)
    assert any(lo <= 40 < hi and hi - lo <= 2 for lo, hi in rg), rg
    assert cost_balanced_ranges(np.zeros(10), 2) == [shard_range(10, 2, r) for r in range(2)]


def test_rccl_id_exchange_through_a_file(tmp_path):
    """The out-of-band leg of mibn_comm_init: rank 0 publishes the 128-byte id atomically, the other ranks of the node
    wait for it (here: a second process that starts waiting before the file exists)."""
    path = str(tmp_path / "id")
    waiter = subprocess.Popen([sys.executable, "-c", (
        "import sys; sys.path.insert(0, %r)\n"
        "from sorobn_amd.sharding import exchange_id\n"
        "uid, _ = exchange_id(1, 2, None, path=%r, timeout_s=60)\n"
        "sys.stdout.write(uid.hex())") % (ROOT, path)], stdout=subprocess.PIPE, text=True)
    import time
    time.sleep(0.5)
    uid = bytes(range(128))
    got, _ = exchange_id(0, 2, lambda: uid, path=path)
    assert got == uid
    out, _ = waiter.communicate(timeout=60)
    assert waiter.returncode == 0 and bytes.fromhex(out) == uid


def test_two_rank_gloo_gather(tmp_path):
    ok = tmp_path / "ok"
    env = dict(os.environ, DIST_OK_FILE=str(ok), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ok.read_text() == "ok 2"


import json  # noqa: E402
import socket  # noqa: E402

import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_launch_path():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per process), on however
    many GPUs the box has: with MIBN_BENCH_BACKEND=files (the dry run, sharding.FileComm) the ranks may share a device and the
    barrier / gather / max-over-ranks go through files - everything but ncclCommInitRank and the collectives themselves.  The ranks
    never import PyTorch (torchrun is only the launcher)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MIBN_BENCH_BACKEND="files", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4096", "--scaling", "weak"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["requests_per_step_per_gpu"] == 4096 and out["config"]["requests_per_step"] == 8192
    assert abs(out["value"] - 2 * 2 * 4096 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert out["per_rank"]["requests_per_step"] == [4096.0, 4096.0]
    assert out["config"]["rccl_ranks"] == 0 and out["config"]["gather"].startswith("DRY RUN")  # no RCCL communicator in the dry run
    assert len(out["per_rank"]["device_pci"]) == 2 and all(":" in p for p in out["per_rank"]["device_pci"])


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher at all: bench.py starts the ranks itself (RANK / WORLD_SIZE / a private
    directory for the RCCL id in their environment).  As a dry run (files) the two ranks may share the box's one GPU."""
    env = dict(os.environ, MIBN_BENCH_BACKEND="files")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096",
           "--global-batch", "10000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong"  # (the default: BASELINE config 4)
    assert out["config"]["requests_per_step"] == 10000 and out["per_rank"]["requests_per_step"] == [5000.0, 5000.0]
    assert abs(out["value"] - 2 * 10000 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


@pytest.mark.gpu
def test_bench_dry_run_of_config_4():
    """BASELINE config 4 as written, as a DRY RUN on however many GPUs the box has (MIBN_BENCH_BACKEND=files, sharding.FileComm):
    `python bench.py --gpus 2 --full-stream` - bench.py spawns the ranks, every rank probes librccl (mibn_comm_probe), they vote,
    rank 0 creates a real RCCL id (ncclGetUniqueId) and the others read it, the first 1 M requests of the stream are split into
    contiguous shards, and ONE gather ends the pass - everything of the N > 1 path except ncclCommInitRank and the collective itself
    (RCCL refuses two ranks on one device).  No PyTorch in the ranks."""
    env = dict(os.environ, MIBN_BENCH_BACKEND="files")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--global-batch", "20000",
           "--full-stream"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(next(l for l in r.stdout.splitlines() if l.startswith('{"metric"')))
    fs = out["full_stream"]
    assert fs["requests"] == 1_000_000 and fs["n_gpus"] == 2 and fs["scaling"] == "strong" and fs["gathers"] == 1
    assert fs["per_rank_requests"] == [500000.0, 500000.0] and abs(fs["posterior_mass"] - 1e6) < 1e-3
    assert fs["imbalance_alg_bytes_max_over_mean"] < 1.05
    echo = [l for l in r.stderr.splitlines() if l.startswith("[mibn comm] rank ")]
    assert len(echo) >= 2 and all("device" in l and "pci" in l for l in echo), r.stderr[-2000:]  # rank / device / links, once per rank


@pytest.mark.gpu
def test_bench_dry_run_of_config_5():
    """BASELINE config 5's launch path rehearsed like config 4's: `python bench.py --gpus 2 --config c5` under
    MIBN_BENCH_BACKEND=files - spawn, librccl probe, vote, rank 0's real ncclGetUniqueId, the id exchange, 2 x 128 chains of ONE Philox
    stream through mibn_gibbs_shard, the int64 histogram reduce onto rank 0 (through files: RCCL refuses two ranks on one device).  The
    pooled histogram is bit for bit the one of the unsharded call with 256 chains, and the estimate agrees with the exact posterior."""
    env = dict(os.environ, MIBN_BENCH_BACKEND="files")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c5", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(next(l for l in r.stdout.splitlines() if l.startswith('{"metric"')))
    assert out["n_gpus"] == 2 and out["chains_total"] == 256 and out["seeds"] == [1000] and out["reduce"].startswith("DRY RUN")
    assert out["max_abs_err_vs_exact"] < 5e-3, out["max_abs_err_vs_exact"]
    import numpy as np
    import netspec
    import sorobn_amd
    bn5 = netspec.build(netspec.grid_spec(5, 10, 8, seed=0), sorobn_amd.BayesNet).use_device(0)
    be = bn5.backend
    rng = np.random.default_rng(1)
    ev5 = {f"{k:03d}": int(rng.integers(0, 8)) for k in (0, 9, 40, 49, 22)}
    q, evs, codes = be.encode(("025",), ev5)
    cycle = sorted([v for v in range(len(be.flat.names)) if v not in set(evs)], key=lambda v: be.flat.names[v])
    whole = be.engine.gibbs(q, evs, codes, 256, 100_000, seed=1000, cycle=cycle)
    assert [int(x) for x in whole] == out["histogram"]
    echo = [l for l in r.stderr.splitlines() if l.startswith("[mibn comm] rank ")]
    assert len(echo) >= 2, r.stderr[-2000:]


def test_collective_vote_without_a_communicator(tmp_path):
    """sharding.all_agree: the ranks of a launch agree on a boolean through marker files (it decides whether a communicator
    can be built at all): one rank's failure is everybody's verdict, and a stale marker of an earlier launch - same tag,
    written before this launch's parent process started - is ignored."""
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from sorobn_amd.sharding import all_agree\n"
            "r = int(sys.argv[1])\n"
            "a, fa = all_agree(r, 2, True, 'first', timeout_s=30)\n"
            "b, fb = all_agree(r, 2, r == 0, 'second', timeout_s=30)\n"
            # a second vote of the same name in the same launch (a second communicator): its own attempt, not the first one's files
            "c, fc = all_agree(r, 2, r == 1, 'first', timeout_s=30)\n"
            "assert fa != fc and os.path.basename(fa).startswith('mibn_vote_t1_first_a1') and '_a2' in os.path.basename(fc)\n"
            "sys.stdout.write('%%d %%d %%d' %% (a, b, c))") % ROOT
    env = dict(os.environ, MIBN_COMM_DIR=str(tmp_path), MIBN_LAUNCH_NONCE="t1")
    stale = tmp_path / "mibn_vote_t1_first_a1.1"
    stale.write_text("0")
    os.utime(stale, (1.0, 1.0))  # 1970: older than any launcher
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert outs == ["1 0 0", "1 0 0"], outs


def test_file_comm_collectives(tmp_path):
    """sharding.FileComm (the transport of the N > 1 dry run): all-gather, int64 reduce, max and barrier between two processes."""
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from sorobn_amd.sharding import FileComm, gather_posteriors\n"
            "r = int(sys.argv[1])\n"
            "c = FileComm(None, r, 2, timeout_s=60)\n"
            "g = c.allgather(np.full((3, 2), float(r)))\n"
            "assert g.shape == (2, 3, 2) and g[0].max() == 0 and g[1].min() == 1\n"
            "full = gather_posteriors(np.arange(4.0 * (2 + r)).reshape(2 + r, 4) + 100 * r, 5, c, ranges=[(0, 2), (2, 5)])\n"
            "assert full.shape == (5, 4) and full[1, 3] == 7 and full[2, 0] == 100 and full[4, 3] == 111\n"
            "h = c.reduce_i64(np.array([1, 2 + r], np.int64))\n"
            "assert h.tolist() == ([2, 5] if r == 0 else [1, 3])\n"
            "assert c.allreduce_max([float(r), 5.0 - r]).tolist() == [1.0, 5.0]\n"
            "for _ in range(5): c.barrier()\n"
            "c.close()\n"
            "sys.stdout.write('ok')") % ROOT
    env = dict(os.environ, MIBN_COMM_DIR=str(tmp_path), MIBN_LAUNCH_NONCE="t2")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert [o[0] for o in outs] == ["ok", "ok"]
    assert len([f for f in os.listdir(tmp_path) if f.startswith("mibn_filecomm")]) <= 2, os.listdir(tmp_path)  # (the last barrier's)


def test_no_pytorch_and_no_second_transport_in_the_product_path():
    """the first N > 1 run must be unable to lie.  bench.py and the package never import PyTorch (the gloo
    stand-in lives in tests/torchcomm.py), bench.py knows exactly two transports - mibn_comm_* ("rccl") and the dry run ("files") -
    and an unknown or failing one ends the rank with a non-zero exit instead of a quiet substitute."""
    import ast
    import glob
    for path in [os.path.join(ROOT, "bench.py"), *glob.glob(os.path.join(ROOT, "sorobn_amd", "*.py"))]:
        tree = ast.parse(open(path).read())
        mods = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                mods.update(a.name.split(".")[0] for a in node.names)
            elif isinstance(node, ast.ImportFrom) and node.module:
                mods.add(node.module.split(".")[0])
        assert "torch" not in mods, path
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "nccl-fallback" not in src and "TorchComm" not in src and "init_process_group" not in src
    from sorobn_amd import sharding
    assert not hasattr(sharding, "TorchComm")
    sys.path.insert(0, ROOT)
    import bench
    with pytest.raises(SystemExit):
        bench.make_comm("gloo", 2, 0, 0, None)

    import inspect
    assert "except" not in inspect.getsource(bench.make_comm)  # nothing between RcclComm's exception and the rank's exit code

    class NoRccl:  # an engine whose librccl probe fails: RcclComm raises
        planner_only = False

        def comm_probe(self):
            raise RuntimeError("cannot load librccl.so")

        def device_info(self):
            return "device 0 test pci 0000:05:00.0 links: none"

    env_keep = {k: os.environ.get(k) for k in ("MIBN_COMM_DIR", "MIBN_LAUNCH_NONCE")}
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.environ.update(MIBN_COMM_DIR=d, MIBN_LAUNCH_NONCE="t_norccl")
        try:
            with pytest.raises(RuntimeError, match="unavailable on at least one rank"):
                sharding.RcclComm(NoRccl(), 0, 1)
        finally:
            for k, v in env_keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    assert bench.pci_numbers("device 0 X pci 0000:c5:00.0 links:") == [0.0, 197.0, 0.0, 0.0] and bench.pci_numbers("nothing") == [-1.0] * 4


def test_id_exchange_publishes_an_error_marker(tmp_path):
    """if rank 0 cannot create the id, the readers fail at once with its message instead of polling for two minutes."""
    path = str(tmp_path / "id")

    def boom():
        raise RuntimeError("ncclGetUniqueId: unhandled system error")

    with pytest.raises(RuntimeError, match="unhandled system error"):
        exchange_id(0, 2, boom, path=path)
    t0 = __import__("time").time()
    with pytest.raises(RuntimeError, match="rank 0 could not create the RCCL id"):
        exchange_id(1, 2, None, path=path, timeout_s=30)
    assert __import__("time").time() - t0 < 5


def test_projected_8gpu_block_is_reproducible_from_the_line():
    """DESIGN section 7's 8-GPU projection must follow from the driver's JSON - `bench.project_8gpu` interpolates the
    measured few-threads rates at (this box's CPU quota / 8) planning threads and multiplies by eight; checked on a made-up line."""
    sys.path.insert(0, ROOT)
    import bench
    line = {"value": 300_000.0, "configs": {"C3_planner_threads_1": {"queries_per_s": 240_000.0}, "C3_two_planner_threads": {"queries_per_s": 250_000.0},
                                            "C3_planner_threads_4": {"queries_per_s": 260_000.0}}}
    quota = bench.host_cpu_quota()
    p = bench.project_8gpu(line)
    assert p["host_cpu_quota"] == quota and p["planner_threads_per_rank_on_8_gpus_same_quota"] == quota / 8
    t = quota / 8
    pts = {1.0: 240e3, 2.0: 250e3, 4.0: 260e3, max(quota, 5.0): 300e3}
    xs = sorted(pts)
    if t <= xs[0]:
        want = pts[xs[0]] * (t if t < 1 else 1.0)
    elif t >= xs[-1]:
        want = pts[xs[-1]]
    else:
        lo = max(x for x in xs if x <= t)
        hi = min(x for x in xs if x >= t)
        want = pts[lo] if hi == lo else pts[lo] + (pts[hi] - pts[lo]) * (t - lo) / (hi - lo)
    assert abs(p["per_rank_queries_per_s"] - want) < 1e-6 * want
    assert abs(p["node_queries_per_s"] - 8 * want) < 1e-6 * want and abs(p["scaling_vs_this_line"] - 8 * want / 300e3) < 1e-9
    assert "PROJECTION" in p["note"]


def test_gpu_session_script_parses():
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "gpu_session.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "gpu_profile.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
