"""GPU-box probe: two ranks of the RCCL transport (sharding.RcclComm -> mibn_comm_*) on a ONE-GPU box.  The id exchange
through the file and both ranks' ncclCommInitRank calls with the same id go through; RCCL itself then refuses two ranks
on one device ("invalid usage") - an N > 1 run needs one GPU per rank, which only the driver's 8-GPU node has.
(Observed in round 2: exactly that; NCCL_IGNORE_DUPLICATE_GPU-style variables do not exist in RCCL 2.27 and hang.)"""
import os, sys, subprocess, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1:
    rank = int(sys.argv[1])
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from sorobn_amd import _capi, sharding
    eng = _capi.Engine(0)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = "2"
    try:
        comm = sharding.RcclComm(eng, rank, 2)
        got = comm.allgather(np.full((2, 3), float(rank)))
        print("rank", rank, "allgather", got.tolist(), "reduce", comm.reduce_i64(np.array([rank + 1, 10], np.int64)).tolist(),
              "max", comm.allreduce_max([float(rank)]).tolist(), flush=True)
        comm.barrier(); comm.close()
    except Exception as e:
        print("rank", rank, "FAILED:", repr(e)[:400], flush=True)
else:
    env = dict(os.environ, MASTER_PORT="29777", MIBN_COMM_DIR="/tmp")
    for extra in ({},):
        print("env extra", extra, flush=True)
        ps = [subprocess.Popen([sys.executable, __file__, str(r)], env=dict(env, **extra)) for r in range(2)]
        for p in ps:
            try: p.wait(timeout=120)
            except subprocess.TimeoutExpired: p.kill(); print("timeout")
