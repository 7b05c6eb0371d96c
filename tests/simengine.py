"""TEST DOUBLE: runs the product planner's step programs on the CPU (oracle/libplan_sim.so).

Lets `pytest -m "not gpu"` exercise the real host logic - flatten.py, Backend.encode /
posterior_series, BayesNet.query / impute post-processing and the C++ planner - in the GPU-less
build container.  Never imported by the product package.
"""
import copy
import ctypes as C
import os
import subprocess

import numpy as np

from sorobn_amd.bayes_net import Backend, CptWatch
from sorobn_amd.flatten import flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libplan_sim.so"])
        L = C.CDLL(os.path.join(ROOT, "oracle", "libplan_sim.so"))
        i32p, i64p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.plan_sim_query.argtypes = [C.c_int32, i32p, i64p, i32p, i64p, f64p, C.c_int32, i32p,
                                     C.c_int32, i32p, C.c_int32, i32p, i32p, f64p, f64p]
        L.plan_sim_query_batch.argtypes = [C.c_int32, i32p, i64p, i32p, i64p, f64p, C.c_int32, i32p, C.c_int64, i64p, i32p, i64p,
                                           i32p, i32p, i64p, f64p, C.c_int32, C.c_int32, f64p]
        L.plan_sim_error.restype = C.c_char_p
        L.plan_sim_set_small_cells.argtypes = [C.c_int]
        L.plan_sim_set_tiling.argtypes = [C.c_int, C.c_int]
        L.plan_sim_set_fuse.argtypes = [C.c_int]
        L.plan_sim_set_chain.argtypes = [C.c_int]
        L.plan_sim_set_sweep.argtypes = [C.c_int]
        L.plan_sim_set_sweep_min.argtypes = [C.c_int]
        L.plan_sim_set_prune.argtypes = [C.c_int]
        _lib = L
    return _lib


class SimEngine:
    """Same surface as sorobn_amd._capi.Engine for the exact path, executed by plan_sim."""

    def __init__(self, flat, small_cells=1024, tiling=(4096, 0), fuse=1):
        self.small_cells = small_cells  # lower it to force FIBER steps on small networks
        self.tiling = tiling            # (big_iters, tile_h): lower them to force tiled levels on small networks
        self.fuse = fuse                # joint elimination of two variables per FIBER step
        self.prune = 1                  # 0: multiply every CPT (full_joint_dist / predict_proba)
        self.chain = 1                  # CHAIN form (three variables per pass), on by default like the product
        self.sweep_min = 2              # fewest variables of a SWEEP pass (2: pair steps too)
        self.sweep = 5                  # SWEEP form (up to five variables per pass, tile in LDS), on by default like the product
        self.f = flat
        self.card = flat.card
        self.last_stats = None
        self.hints = np.stack(flat.hints).astype(np.int32) if flat.hints else np.zeros((0, len(flat.card)), np.int32)

    def _one(self, q, ev, codes):
        f = self.f
        L = lib()
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        q = np.ascontiguousarray(q, np.int32)
        ev_ = np.ascontiguousarray(ev if len(ev) else [0], np.int32)
        co_ = np.ascontiguousarray(codes if len(codes) else [0], np.int32)
        cells = int(np.prod(f.card[q].astype(np.int64)))
        out = np.zeros(cells, np.float64)
        stats = np.zeros(5, np.float64)
        hints = np.ascontiguousarray(self.hints.reshape(-1) if self.hints.size else [0], np.int32)
        L.plan_sim_set_small_cells(int(self.small_cells))
        L.plan_sim_set_tiling(int(self.tiling[0]), int(self.tiling[1]))
        L.plan_sim_set_fuse(int(self.fuse))
        L.plan_sim_set_prune(int(self.prune))
        L.plan_sim_set_chain(int(self.chain))
        L.plan_sim_set_sweep(int(self.sweep))
        L.plan_sim_set_sweep_min(int(self.sweep_min))
        rc = L.plan_sim_query(len(f.card), p(f.card, C.c_int32), p(f.scope_off, C.c_int64),
                              p(f.scope_vars, C.c_int32), p(f.value_off, C.c_int64),
                              p(f.values, C.c_double), self.hints.shape[0], p(hints, C.c_int32),
                              len(q), p(q, C.c_int32), len(ev), p(ev_, C.c_int32),
                              p(co_, C.c_int32), p(out, C.c_double), p(stats, C.c_double))
        if rc != 0:
            raise RuntimeError(f"plan_sim_query rc={rc}: {L.plan_sim_error().decode()}")
        self.last_stats = stats
        return out

    def batch(self, qvars, evars, ecodes, stagger=1, threads=1):
        """Fixed-shape batch through ONE plan_batch + build_schedule (multi-request levels, `stagger` groups, `threads`
        planning workers), executed level by level like the kernel: posteriors [B, cells]."""
        f = self.f
        L = lib()
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        qvars = np.ascontiguousarray(qvars, np.int32).reshape(len(qvars), -1)
        B, nq = qvars.shape
        evars = np.ascontiguousarray(evars, np.int32).reshape(B, -1)
        ecodes = np.ascontiguousarray(ecodes, np.int32).reshape(B, -1)
        ne = evars.shape[1]
        q_off = np.arange(B + 1, dtype=np.int64) * nq
        e_off = np.arange(B + 1, dtype=np.int64) * ne
        cells = np.prod(f.card[qvars].astype(np.int64), axis=1)
        out_off = np.concatenate([[0], np.cumsum(cells)]).astype(np.int64)
        out = np.zeros(int(out_off[-1]), np.float64)
        stats = np.zeros(5, np.float64)
        hints = np.ascontiguousarray(self.hints.reshape(-1) if self.hints.size else [0], np.int32)
        L.plan_sim_set_small_cells(int(self.small_cells))
        L.plan_sim_set_tiling(int(self.tiling[0]), int(self.tiling[1]))
        L.plan_sim_set_fuse(int(self.fuse))
        L.plan_sim_set_prune(int(self.prune))
        L.plan_sim_set_chain(int(self.chain))
        L.plan_sim_set_sweep(int(self.sweep))
        L.plan_sim_set_sweep_min(int(self.sweep_min))
        ev_ = evars.reshape(-1) if ne else np.zeros(1, np.int32)
        ec_ = ecodes.reshape(-1) if ne else np.zeros(1, np.int32)
        rc = L.plan_sim_query_batch(len(f.card), p(f.card, C.c_int32), p(f.scope_off, C.c_int64), p(f.scope_vars, C.c_int32),
                                    p(f.value_off, C.c_int64), p(f.values, C.c_double), self.hints.shape[0], p(hints, C.c_int32),
                                    B, p(q_off, C.c_int64), p(qvars.reshape(-1), C.c_int32), p(e_off, C.c_int64), p(ev_, C.c_int32),
                                    p(ec_, C.c_int32), p(out_off, C.c_int64), p(out, C.c_double), int(stagger), int(threads),
                                    p(stats, C.c_double))
        if rc != 0:
            raise RuntimeError(f"plan_sim_query_batch rc={rc}: {L.plan_sim_error().decode()}")
        self.last_stats = stats
        return out.reshape(B, -1)

    def gibbs(self, qvars, evars, ecodes, n_chains, n_iterations, seed=0, cycle=None, chain_first=0):
        """STAND-IN (no chain on the CPU): the exact posterior scaled to counts, so that the host side of the Gibbs path -
        `accelerate`'s rebind, Backend.gibbs_sampling, the Series construction - runs in the GPU-less container."""
        exact = self._one(np.asarray(qvars, np.int32), np.asarray(evars, np.int32), np.asarray(ecodes, np.int32))
        total = int(n_chains) * int(n_iterations)
        counts = np.floor(exact * total).astype(np.int64)
        counts[int(np.argmax(exact))] += total - int(counts.sum())
        return counts

    def set_option(self, name, value):
        if name == "prune":
            self.prune = int(value)
        elif name == "chain":
            self.chain = int(value)
        elif name == "sweep":
            self.sweep = int(value)
        elif name == "sweep_min":
            self.sweep_min = int(value)
        elif name == "tiny":
            pass  # (the small-network kernel exists on the device only)
        else:
            raise KeyError(name)

    def _with_flags(self, flags, fn):
        """MIBN_Q_NOPRUNE (per call, include/mibn.h): every CPT takes part."""
        saved = self.prune
        if flags & 1:
            self.prune = 0
        try:
            return fn()
        finally:
            self.prune = saved

    def query_fixed(self, qvars, evars, ecodes, flags=0):
        return self._with_flags(flags, lambda: np.stack([self._one(q, e, c) for q, e, c in zip(qvars, evars, ecodes)]))

    def query_batch(self, q_off, q_vars, e_off, e_vars, e_codes, out_off=None, flags=0):
        outs = self._with_flags(flags, lambda: [self._one(q_vars[a:b], e_vars[c:d], e_codes[c:d])
                                                for a, b, c, d in zip(q_off[:-1], q_off[1:], e_off[:-1], e_off[1:])])
        off = np.concatenate([[0], np.cumsum([len(o) for o in outs])]).astype(np.int64)
        return (np.concatenate(outs) if outs else np.zeros(0)), off


def sim_backend(bn, small_cells=1024, tiling=(4096, 0), fuse=1):
    """A Backend whose engine is the CPU plan simulator (bypasses Backend.__init__)."""
    b = Backend.__new__(Backend)
    b.flat = flatten(bn)
    b.watch = CptWatch(bn)
    b.engine = SimEngine(b.flat, small_cells, tiling, fuse)
    b._anc = {}
    # the presence network (full_joint_dist(keep_zeros=True)): same structure, tables = 1.0 where a CPT row exists
    pf = copy.copy(b.flat)
    pf.values = b.flat.present
    b._presence = SimEngine(pf, small_cells, tiling, fuse)
    b._device, b._planner_only = None, True
    return b


def attach(bn, small_cells=1024, tiling=(4096, 0), fuse=1):
    """Make a sorobn_amd.BayesNet answer through the simulator (tests only)."""
    bn._backend = sim_backend(bn, small_cells, tiling, fuse)
    return bn
