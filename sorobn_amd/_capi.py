"""ctypes binding of the C-ABI in include/mibn.h (libmibn.so, built in-tree by __graft_entry__.build()).

There is deliberately no fallback: if the library is missing or no gfx950 device is visible the
calls raise, they never route to a CPU implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIBN_LIB") or os.path.join(_HERE, "libmibn.so")  # MIBN_LIB: kernel-variant experiments

# every symbol include/mibn.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "mibn_device_count", "mibn_version", "mibn_create", "mibn_destroy", "mibn_last_error",
    "mibn_set_network", "mibn_set_order_hints", "mibn_query_batch", "mibn_last_stats",
    "mibn_plan_stats", "mibn_create_planner", "mibn_set_option", "mibn_gibbs",
    "mibn_last_kernel_stats", "mibn_submit_batch", "mibn_wait", "mibn_drain", "mibn_total_stats",
    "mibn_total_kernel_stats", "mibn_sample", "mibn_sampling_query", "mibn_count_tables",
    "mibn_query_batch_ex", "mibn_plan_order", "mibn_estimate_costs", "mibn_device_synchronize", "mibn_gibbs_shard",
    "mibn_comm_unique_id", "mibn_comm_init", "mibn_comm_destroy", "mibn_comm_allgather_f64",
    "mibn_comm_reduce_i64", "mibn_comm_allreduce_max_f64", "mibn_comm_barrier", "mibn_gibbs_conditional", "mibn_sample_probe",
    "mibn_comm_probe", "mibn_device_info", "mibn_comm_count",
]

OK, E_ARG, E_NODEVICE, E_HIP, E_NOMEM, E_STATE, E_LIMIT, E_COMM = 0, -1, -2, -3, -4, -5, -6, -7
Q_NOPRUNE = 1
COMM_ID_BYTES = 128


class MibnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mibn error {code}: {msg}")
        self.code = code
        self.msg = msg


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "alg_bytes", "alg_flops", "n_steps", "kernel_ms", "plan_ms", "h2d_ms", "d2h_ms",
        "total_ms", "n_launches", "arena_bytes", "max_step_cells", "n_workgroups")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_double), ("ms", C.c_double),
                ("alg_bytes", C.c_double), ("items", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        i32p, i64p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        vp = C.c_void_p
        L.mibn_device_count.argtypes = [C.POINTER(C.c_int)]
        L.mibn_version.restype = C.c_char_p
        L.mibn_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.mibn_create_planner.argtypes = [C.POINTER(vp)]
        L.mibn_destroy.argtypes = [vp]
        L.mibn_destroy.restype = None
        L.mibn_last_error.argtypes = [vp]
        L.mibn_last_error.restype = C.c_char_p
        L.mibn_set_network.argtypes = [vp, C.c_int32, i32p, i64p, i32p, i64p, f64p]
        L.mibn_set_order_hints.argtypes = [vp, C.c_int32, i32p]
        L.mibn_query_batch.argtypes = [vp, C.c_int64, i64p, i32p, i64p, i32p, i32p, i64p, f64p]
        L.mibn_query_batch_ex.argtypes = [vp, C.c_uint32, C.c_int64, i64p, i32p, i64p, i32p, i32p, i64p, f64p]
        L.mibn_plan_order.argtypes = [vp, C.c_int32, i32p, C.c_int32, i32p, i32p, i32p]
        L.mibn_estimate_costs.argtypes = [vp, C.c_int64, i64p, i32p, i64p, i32p, f64p]
        L.mibn_device_synchronize.argtypes = [vp]
        L.mibn_gibbs_shard.argtypes = [vp, C.c_int32, i32p, C.c_int32, i32p, i32p, i32p, C.c_int64, C.c_int64,
                                       C.c_int64, C.c_uint64, i64p]
        L.mibn_gibbs_conditional.argtypes = [vp, C.c_int32, i32p, i32p, i32p, C.c_int32, C.c_int64, C.POINTER(C.c_uint8), f64p]
        L.mibn_comm_probe.argtypes = [vp]
        L.mibn_device_info.argtypes = [vp, C.c_char_p, C.c_int32]
        L.mibn_comm_unique_id.argtypes = [vp, C.c_char_p]
        L.mibn_comm_init.argtypes = [vp, C.c_int32, C.c_int32, C.c_char_p]
        L.mibn_comm_destroy.argtypes = [vp]
        L.mibn_comm_count.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mibn_comm_allgather_f64.argtypes = [vp, f64p, C.c_int64, f64p]
        L.mibn_comm_reduce_i64.argtypes = [vp, i64p, C.c_int64, C.c_int32]
        L.mibn_comm_allreduce_max_f64.argtypes = [vp, f64p, C.c_int64]
        L.mibn_comm_barrier.argtypes = [vp]
        L.mibn_submit_batch.argtypes = [vp, C.c_int64, i64p, i32p, i64p, i32p, i32p, i64p, f64p, C.POINTER(C.c_int32)]
        L.mibn_wait.argtypes = [vp, C.c_int32]
        L.mibn_drain.argtypes = [vp]
        L.mibn_total_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mibn_total_kernel_stats.argtypes = [vp, C.c_int32, C.POINTER(KernelStat), C.POINTER(C.c_int32)]
        L.mibn_sample.argtypes = [vp, C.c_int64, C.c_int32, i32p, i32p, C.c_uint64, C.POINTER(C.c_uint8)]
        L.mibn_sample_probe.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint8), C.c_int32, f64p, f64p]
        L.mibn_sampling_query.argtypes = [vp, C.c_int32, C.c_int32, i32p, C.c_int32, i32p, i32p, C.c_int64, C.c_uint64, f64p, i64p]
        L.mibn_count_tables.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_uint8), C.c_int32, i32p, C.c_int32, i64p, i32p, i64p, i64p]
        L.mibn_last_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mibn_last_kernel_stats.argtypes = [vp, C.c_int32, C.POINTER(KernelStat), C.POINTER(C.c_int32)]
        L.mibn_plan_stats.argtypes = [vp, C.c_int32, i32p, C.c_int32, i32p, C.POINTER(Stats)]
        L.mibn_set_option.argtypes = [vp, C.c_char_p, C.c_double]
        L.mibn_gibbs.argtypes = [vp, C.c_int32, i32p, C.c_int32, i32p, i32p, i32p, C.c_int64,
                                 C.c_int64, C.c_uint64, i64p]
        _lib = L
    return _lib


def device_count():
    n = C.c_int(0)
    lib().mibn_device_count(C.byref(n))
    return n.value


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Engine:
    """One mibn context (= one GPU) holding one flattened network."""

    def __init__(self, device=0, planner_only=False):
        self._h = C.c_void_p()
        self._L = lib()
        if planner_only:
            rc = self._L.mibn_create_planner(C.byref(self._h))
        else:
            rc = self._L.mibn_create(int(device), C.byref(self._h))
        if rc != OK:
            why = {E_NODEVICE: "no gfx950 HIP device visible (there is no CPU fallback)",
                   E_ARG: "bad device index", E_HIP: "HIP runtime error"}.get(rc, "?")
            raise MibnError(rc, f"mibn_create failed: {why}")
        self.planner_only = planner_only
        self.device = device
        # experiment hook: MIBN_OPTS="name=value,..." sets engine options on every engine of the process (a parity run of the GPU
        # suite under a non-default option, e.g. MIBN_OPTS=mfma_kernel=1 python -m pytest tests -m gpu)
        for kv in os.environ.get("MIBN_OPTS", "").split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                self.set_option(k.strip(), float(v))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.mibn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise MibnError(rc, self._L.mibn_last_error(self._h).decode())

    def set_network(self, card, scope_off, scope_vars, value_off, values):
        card, scope_vars = _i32(card), _i32(scope_vars)
        scope_off, value_off = _i64(scope_off), _i64(value_off)
        values = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._L.mibn_set_network(
            self._h, len(card), _p(card, C.c_int32), _p(scope_off, C.c_int64),
            _p(scope_vars, C.c_int32), _p(value_off, C.c_int64), _p(values, C.c_double)))
        self.card = card
        self._card_list = card.tolist()
        self._one = {}

    def set_order_hints(self, hints):
        h = _i32(np.asarray(hints).reshape(-1, len(self.card)))
        self._check(self._L.mibn_set_order_hints(self._h, h.shape[0], _p(h, C.c_int32)))

    def set_option(self, name, value):
        self._check(self._L.mibn_set_option(self._h, name.encode(), float(value)))

    def query_batch(self, q_off, q_vars, e_off, e_vars, e_codes, out_off=None, flags=0):
        """CSR request batch -> flat float64 posteriors (+ out_off).  flags: Q_NOPRUNE (per call, see mibn.h)."""
        q_off, e_off = _i64(q_off), _i64(e_off)
        q_vars, e_vars, e_codes = _i32(q_vars), _i32(e_vars), _i32(e_codes)
        B = len(q_off) - 1
        if out_off is None:
            cells = np.ones(B, np.int64)
            if B:
                nq = np.diff(q_off)
                if nq.min() == nq.max() and nq[0] > 0:
                    cells = np.prod(self.card[q_vars].reshape(B, -1).astype(np.int64), axis=1)
                else:
                    cells = np.array([int(np.prod(self.card[q_vars[a:b]].astype(np.int64)))
                                      for a, b in zip(q_off[:-1], q_off[1:])], np.int64)
            out_off = np.concatenate([[0], np.cumsum(cells)]).astype(np.int64)
        out_off = _i64(out_off)
        out = np.zeros(int(out_off[-1]), np.float64)
        e_vars_ = e_vars if len(e_vars) else np.zeros(1, np.int32)
        e_codes_ = e_codes if len(e_codes) else np.zeros(1, np.int32)
        q_vars_ = q_vars if len(q_vars) else np.zeros(1, np.int32)
        out_ = out if len(out) else np.zeros(1, np.float64)
        self._check(self._L.mibn_query_batch_ex(
            self._h, int(flags), B, _p(q_off, C.c_int64), _p(q_vars_, C.c_int32), _p(e_off, C.c_int64),
            _p(e_vars_, C.c_int32), _p(e_codes_, C.c_int32), _p(out_off, C.c_int64),
            _p(out_, C.c_double)))
        return out, out_off

    def query_one(self, q, ev, codes, flags=0):
        """One request (lists of variable ids / codes) -> dense posterior [cells].  The latency path of a single `query()`:
        preallocated ctypes arrays per request shape, one C call (query_fixed builds a dozen numpy arrays on the way)."""
        nq, ne = len(q), len(ev)
        card = self._card_list
        cells = 1
        try:
            for v in q:
                cells *= card[v]
        except (IndexError, TypeError):
            cells = 1  # (an unknown id: the library builds the error message)
        one = self._one
        key = (nq, ne)
        buf = one.get(key)
        if buf is None:
            buf = one[key] = ((C.c_int64 * 2)(0, nq), (C.c_int64 * 2)(0, ne), (C.c_int64 * 2)(0, 0),
                              (C.c_int32 * max(1, nq))(), (C.c_int32 * max(1, ne))(), (C.c_int32 * max(1, ne))())
        q_off, e_off, out_off, qa, ea, ca = buf
        out_off[1] = cells
        qa[:nq] = q
        if ne:
            ea[:ne] = ev
            ca[:ne] = codes
        out = np.zeros(max(1, cells), np.float64)
        rc = self._L.mibn_query_batch_ex(self._h, int(flags), 1, q_off, qa, e_off, ea, ca, out_off,
                                         out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != OK:
            self._check(rc)
        return out[:cells]

    def query_fixed(self, qvars, evars, ecodes, flags=0):
        """Fixed-shape batch: qvars[B, nq], evars[B, ne], ecodes[B, ne] -> posteriors[B, cells]
        (all requests must have the same query-table size)."""
        if len(qvars) == 0:
            return np.zeros((0, 0), np.float64)
        qvars = _i32(qvars).reshape(len(qvars), -1)
        B, nq = qvars.shape
        evars = _i32(evars).reshape(B, -1)
        ecodes = _i32(ecodes).reshape(B, -1)
        ne = evars.shape[1]
        q_off = np.arange(B + 1, dtype=np.int64) * nq
        e_off = np.arange(B + 1, dtype=np.int64) * ne
        out, out_off = self.query_batch(q_off, qvars.reshape(-1), e_off, evars.reshape(-1),
                                        ecodes.reshape(-1), flags=flags)
        return out.reshape(B, -1) if B else out.reshape(0, 0)

    def submit_fixed(self, qvars, evars, ecodes):
        """Asynchronous query_fixed: returns a handle for wait().  At most two calls in flight."""
        qvars = _i32(qvars).reshape(len(qvars), -1)
        B, nq = qvars.shape
        evars = _i32(evars).reshape(B, -1)
        ecodes = _i32(ecodes).reshape(B, -1)
        ne = evars.shape[1]
        q_off = np.arange(B + 1, dtype=np.int64) * nq
        e_off = np.arange(B + 1, dtype=np.int64) * ne
        cells = np.prod(self.card[qvars].astype(np.int64), axis=1) if B else np.zeros(0, np.int64)
        out_off = np.concatenate([[0], np.cumsum(cells)]).astype(np.int64)
        out = np.zeros(max(1, int(out_off[-1])), np.float64)
        ticket = C.c_int32(-1)
        keep = (q_off, qvars, e_off, evars, ecodes, out_off, out)  # the library reads them during the call only, `out` until wait
        self._check(self._L.mibn_submit_batch(
            self._h, B, _p(q_off, C.c_int64), _p(qvars.reshape(-1), C.c_int32), _p(e_off, C.c_int64),
            _p(evars.reshape(-1) if ne else np.zeros(1, np.int32), C.c_int32),
            _p(ecodes.reshape(-1) if ne else np.zeros(1, np.int32), C.c_int32), _p(out_off, C.c_int64),
            _p(out, C.c_double), C.byref(ticket)))
        return {"ticket": ticket.value, "out": out, "B": B, "n": int(out_off[-1]), "keep": keep}

    def wait(self, handle):
        self._check(self._L.mibn_wait(self._h, handle["ticket"]))
        return handle["out"][:handle["n"]].reshape(handle["B"], -1) if handle["B"] else handle["out"][:0].reshape(0, 0)

    def wait_flat(self, handle):
        """wait() for a call whose query tables differ in size from request to request (a 2-state and a 3-state query
        variable in one fixed-arity batch): the posteriors back to back, request i at the offsets the caller derives from the cards."""
        self._check(self._L.mibn_wait(self._h, handle["ticket"]))
        return handle["out"][:handle["n"]]

    def drain(self):
        self._check(self._L.mibn_drain(self._h))

    def stats(self):
        s = Stats()
        self._check(self._L.mibn_last_stats(self._h, C.byref(s)))
        return s.as_dict()

    def total_stats(self):
        """Counters accumulated since the engine was created (take differences over a region of calls)."""
        s = Stats()
        self._check(self._L.mibn_total_stats(self._h, C.byref(s)))
        return s.as_dict()

    def total_kernel_stats(self):
        arr = (KernelStat * 64)()
        n = C.c_int32(0)
        self._check(self._L.mibn_total_kernel_stats(self._h, 64, arr, C.byref(n)))
        return {arr[i].name.decode(): {"launches": arr[i].launches, "ms": arr[i].ms, "alg_bytes": arr[i].alg_bytes,
                                       "items": arr[i].items} for i in range(n.value)}

    def kernel_stats(self):
        """Per-kernel breakdown of the last query_batch: list of dicts (name, launches, ms, alg_bytes, items)."""
        arr = (KernelStat * 64)()
        n = C.c_int32(0)
        self._check(self._L.mibn_last_kernel_stats(self._h, 64, arr, C.byref(n)))
        return [{"name": arr[i].name.decode(), "launches": arr[i].launches, "ms": arr[i].ms,
                 "alg_bytes": arr[i].alg_bytes, "items": arr[i].items} for i in range(n.value)]

    def plan_stats(self, qvars, evars):
        q, e = _i32(qvars), _i32(evars)
        e_ = e if len(e) else np.zeros(1, np.int32)
        s = Stats()
        self._check(self._L.mibn_plan_stats(self._h, len(q), _p(q, C.c_int32), len(e),
                                            _p(e_, C.c_int32), C.byref(s)))
        return s.as_dict()

    def plan_order(self, qvars, evars):
        """The elimination order the planner executes for this request (hidden variables, first eliminated first)."""
        q, e = _i32(qvars), _i32(evars)
        e_ = e if len(e) else np.zeros(1, np.int32)
        order = np.zeros(len(self.card), np.int32)
        n = C.c_int32(0)
        self._check(self._L.mibn_plan_order(self._h, len(q), _p(q, C.c_int32), len(e), _p(e_, C.c_int32),
                                            _p(order, C.c_int32), C.byref(n)))
        return order[:n.value].copy()

    def gibbs(self, qvars, evars, ecodes, n_chains, n_iterations, seed=0, cycle=None, chain_first=0):
        """Histogram of chains [chain_first, chain_first + n_chains) of stream `seed` (mibn_gibbs_shard)."""
        q, e, c = _i32(qvars), _i32(evars), _i32(ecodes)
        e_ = e if len(e) else np.zeros(1, np.int32)
        c_ = c if len(c) else np.zeros(1, np.int32)
        cells = int(np.prod(self.card[q].astype(np.int64)))
        counts = np.zeros(cells, np.int64)
        cyc = None
        if cycle is not None:
            cyc_arr = _i32(cycle)
            cyc = _p(cyc_arr, C.c_int32)
        self._check(self._L.mibn_gibbs_shard(self._h, len(q), _p(q, C.c_int32), len(e),
                                             _p(e_, C.c_int32), _p(c_, C.c_int32), cyc, int(chain_first), int(n_chains),
                                             int(n_iterations), int(seed) & (2**64 - 1),
                                             _p(counts, C.c_int64)))
        return counts

    def gibbs_conditional(self, var, states, evars=(), ecodes=(), cycle=None):
        """P(var = x | the rest of each row of `states` [n_rows, n_vars]) as the Gibbs kernel computes it (mibn_gibbs_conditional)."""
        e, c = _i32(evars), _i32(ecodes)
        e_ = e if len(e) else np.zeros(1, np.int32)
        c_ = c if len(c) else np.zeros(1, np.int32)
        st = np.ascontiguousarray(states, np.uint8).reshape(-1, len(self.card))
        out = np.zeros((len(st), int(self.card[var])), np.float64)
        cyc = None
        if cycle is not None:
            cyc_arr = _i32(cycle)
            cyc = _p(cyc_arr, C.c_int32)
        self._check(self._L.mibn_gibbs_conditional(self._h, len(e), _p(e_, C.c_int32), _p(c_, C.c_int32), cyc, int(var), len(st),
                                                   st.ctypes.data_as(C.POINTER(C.c_uint8)), _p(out, C.c_double)))
        return out

    # ---- shard balancing / multi-GPU ------------------------------------------------------
    def estimate_costs(self, qvars, evars):
        """Planner cost estimate (section-8(d) bytes of the cheaper sweep order) per request of a fixed-shape batch."""
        qvars = _i32(qvars).reshape(len(qvars), -1)
        B, nq = qvars.shape
        evars = _i32(evars).reshape(B, -1)
        ne = evars.shape[1]
        q_off = np.arange(B + 1, dtype=np.int64) * nq
        e_off = np.arange(B + 1, dtype=np.int64) * ne
        cost = np.zeros(max(1, B), np.float64)
        ev = evars.reshape(-1) if evars.size else np.zeros(1, np.int32)
        self._check(self._L.mibn_estimate_costs(self._h, B, _p(q_off, C.c_int64), _p(qvars.reshape(-1), C.c_int32),
                                                _p(e_off, C.c_int64), _p(ev, C.c_int32), _p(cost, C.c_double)))
        return cost[:B]

    def synchronize(self):
        self._check(self._L.mibn_device_synchronize(self._h))

    def comm_probe(self):
        """librccl.so loads and has the entry points mibn_comm_* needs (no id, no socket, no thread)."""
        self._check(self._L.mibn_comm_probe(self._h))

    def device_info(self):
        buf = C.create_string_buffer(1024)
        self._check(self._L.mibn_device_info(self._h, buf, 1024))
        return buf.value.decode(errors="replace")

    def comm_unique_id(self):
        buf = C.create_string_buffer(COMM_ID_BYTES)
        self._check(self._L.mibn_comm_unique_id(self._h, buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self._check(self._L.mibn_comm_init(self._h, int(rank), int(world), unique_id))

    def comm_count(self):
        """(ncclCommCount, ncclCommUserRank) of the live communicator - what RCCL itself reports."""
        n, r = C.c_int32(0), C.c_int32(-1)
        self._check(self._L.mibn_comm_count(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def comm_destroy(self):
        self._check(self._L.mibn_comm_destroy(self._h))

    def comm_allgather(self, send, world):
        send = np.ascontiguousarray(send, dtype=np.float64).reshape(-1)
        recv = np.empty(len(send) * int(world), np.float64)
        self._check(self._L.mibn_comm_allgather_f64(self._h, _p(send, C.c_double), len(send), _p(recv, C.c_double)))
        return recv.reshape(int(world), -1)

    def comm_reduce_i64(self, buf, root=0):
        out = np.ascontiguousarray(buf, dtype=np.int64).copy()
        self._check(self._L.mibn_comm_reduce_i64(self._h, _p(out.reshape(-1), C.c_int64), out.size, int(root)))
        return out

    def comm_allreduce_max(self, values):
        out = np.ascontiguousarray(values, dtype=np.float64).copy().reshape(-1)
        self._check(self._L.mibn_comm_allreduce_max_f64(self._h, _p(out, C.c_double), len(out)))
        return out

    def comm_barrier(self):
        self._check(self._L.mibn_comm_barrier(self._h))

    def sample(self, n_samples, init_vars=(), init_codes=(), seed=0):
        """Forward samples: uint8 codes [n_samples, n_vars]."""
        iv, ic = _i32(init_vars), _i32(init_codes)
        out = np.zeros((int(n_samples), len(self.card)), np.uint8)
        iv_ = iv if len(iv) else np.zeros(1, np.int32)
        ic_ = ic if len(ic) else np.zeros(1, np.int32)
        self._check(self._L.mibn_sample(self._h, int(n_samples), len(iv), _p(iv_, C.c_int32), _p(ic_, C.c_int32),
                                        int(seed) & (2**64 - 1), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def sample_probe(self, states):
        """The sampling kernel's walk over GIVEN joint states [n_rows, n_vars] (mibn_sample_probe) -> (likelihood [n_rows],
        running sums of every variable's conditional row [n_rows, n_vars, max card])."""
        st = np.ascontiguousarray(states, np.uint8).reshape(-1, len(self.card))
        stride = int(self.card.max())
        lik = np.zeros(len(st), np.float64)
        cdf = np.zeros((len(st), len(self.card), stride), np.float64)
        self._check(self._L.mibn_sample_probe(self._h, len(st), st.ctypes.data_as(C.POINTER(C.c_uint8)), stride,
                                              _p(lik, C.c_double), _p(cdf, C.c_double)))
        return lik, cdf

    def sampling_query(self, mode, qvars, evars, ecodes, n_samples, seed=0):
        """mode 1 = rejection, 2 = likelihood weighting -> (weight_sum, counts) per joint query state."""
        q, e, c = _i32(qvars), _i32(evars), _i32(ecodes)
        e_ = e if len(e) else np.zeros(1, np.int32)
        c_ = c if len(c) else np.zeros(1, np.int32)
        cells = int(np.prod(self.card[q].astype(np.int64)))
        wsum = np.zeros(cells, np.float64)
        counts = np.zeros(cells, np.int64)
        self._check(self._L.mibn_sampling_query(self._h, int(mode), len(q), _p(q, C.c_int32), len(e), _p(e_, C.c_int32),
                                                _p(c_, C.c_int32), int(n_samples), int(seed) & (2**64 - 1),
                                                _p(wsum, C.c_double), _p(counts, C.c_int64)))
        return wsum, counts

    def count_tables(self, codes, card, tables):
        """codes: uint8 [n_rows, n_cols] (row- or column-major, sent as it is); tables: list of column-index tuples
        -> list of dense int64 contingency tables shaped by the cards of their columns."""
        codes = np.asarray(codes, dtype=np.uint8)
        if codes.ndim != 2:
            raise ValueError("codes must be a [n_rows, n_cols] matrix")
        if not (codes.flags.f_contiguous or codes.flags.c_contiguous):
            codes = np.ascontiguousarray(codes)
        row_major = 0 if codes.flags.f_contiguous else 1
        n_rows, n_cols = codes.shape
        card = _i32(card)
        scope_off = np.concatenate([[0], np.cumsum([len(t) for t in tables])]).astype(np.int64)
        scope_cols = _i32([c for t in tables for c in t]) if len(tables) else np.zeros(0, np.int32)
        cells = [int(np.prod(card[list(t)].astype(np.int64))) for t in tables]
        counts_off = np.concatenate([[0], np.cumsum(cells)]).astype(np.int64)
        counts = np.zeros(max(1, int(counts_off[-1])), np.int64)
        sc = scope_cols if len(scope_cols) else np.zeros(1, np.int32)
        self._check(self._L.mibn_count_tables(
            self._h, n_rows, n_cols, codes.ctypes.data_as(C.POINTER(C.c_uint8)), row_major, _p(card if len(card) else np.zeros(1, np.int32), C.c_int32),
            len(tables), _p(scope_off, C.c_int64), _p(sc, C.c_int32), _p(counts_off, C.c_int64), _p(counts, C.c_int64)))
        return [counts[a:b].reshape([int(card[c]) for c in t]) for t, a, b in zip(tables, counts_off[:-1], counts_off[1:])]
