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
    many GPUs the box has: with the MIBN_BENCH_BACKEND=gloo test hook the ranks may share a device and the barrier /
    gather / max-over-ranks run on host tensors - everything but the RCCL transport itself."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MIBN_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4096"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["requests_per_step_per_gpu"] == 4096
    assert abs(out["value"] - 2 * 2 * 4096 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher at all: bench.py starts the ranks itself (RANK / WORLD_SIZE / a private
    directory for the RCCL id in their environment).  With the gloo test hook the two ranks may share the box's one GPU."""
    env = dict(os.environ, MIBN_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2
    assert abs(out["value"] - 2 * 2 * 4096 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_collective_vote_without_a_communicator(tmp_path):
    """sharding.all_agree: the ranks of a launch agree on a boolean through marker files (it decides whether a communicator
    can be built at all): one rank's failure is everybody's verdict, and a stale marker of an earlier launch - same tag,
    written before this launch's parent process started - is ignored."""
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from sorobn_amd.sharding import all_agree\n"
            "r = int(sys.argv[1])\n"
            "a = all_agree(r, 2, True, 'first', timeout_s=30)\n"
            "b = all_agree(r, 2, r == 0, 'second', timeout_s=30)\n"
            "sys.stdout.write('%%d %%d' %% (a, b))") % ROOT
    env = dict(os.environ, MIBN_COMM_DIR=str(tmp_path), MIBN_LAUNCH_NONCE="t1")
    stale = tmp_path / "mibn_vote_t1_first.1"
    stale.write_text("0")
    os.utime(stale, (1.0, 1.0))  # 1970: older than any launcher
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert outs == ["1 0", "1 0"], outs
