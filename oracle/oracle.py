"""TEST INFRASTRUCTURE - ctypes front-end of the C oracle (oracle/ve_oracle.c).

Converts a network *spec* (tests/netspec.py) into the sparse code-row tables the oracle consumes,
independently of sorobn_amd's flattening, and answers queries by the restated reference algorithm.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libve_oracle.so")
    src = os.path.join(_HERE, "ve_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libve_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.ve_net_create.restype = C.c_void_p
        L.ve_net_create.argtypes = [C.c_int, i32p]
        L.ve_net_destroy.argtypes = [C.c_void_p]
        L.ve_net_set_cpt.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, C.c_int64, i32p, f64p]
        L.ve_query.restype = C.c_int64
        L.ve_query.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int, i32p, i32p, i32p, C.c_int64,
                               i32p, f64p]
        L.ve_last_stats.argtypes = [C.c_void_p, f64p, f64p]
        L.ve_pointwise_mul_two.restype = C.c_int64
        L.ve_pointwise_mul_two.argtypes = [i32p, C.c_int, i32p, C.c_int64, i32p, f64p, C.c_int,
                                           i32p, C.c_int64, i32p, f64p, C.c_int64, i32p, i32p,
                                           i32p, f64p]
        L.ve_sum_out.restype = C.c_int64
        L.ve_sum_out.argtypes = [i32p, C.c_int, i32p, C.c_int64, i32p, f64p, C.c_int32, C.c_int64,
                                 i32p, i32p, i32p, f64p]
        _LIB = L
    return _LIB


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _code_of(dom, label):
    """Evidence labels match by Python equality (factor.index.get_level_values(var) == val,
    bayes_net.py:774): 1 matches True.  Unknown label -> -1 (matches nothing)."""
    for i, d in enumerate(dom):
        try:
            if d == label:
                return i
        except Exception:
            pass
    return -1


class OracleNet:
    """Sparse-table network for the C oracle, built from a netspec spec."""

    def __init__(self, spec):
        self.spec = spec
        self.names = list(spec["nodes"])
        self.id = {n: i for i, n in enumerate(self.names)}
        dom = {n: set() for n in self.names}
        for cpt in spec["cpts"].values():
            for r in cpt["rows"]:
                for nme, lab in zip(cpt["names"], r[:-1]):
                    dom[nme].add(lab)
        self.dom = {n: sorted(v) for n, v in dom.items()}
        self.card = np.array([len(self.dom[n]) for n in self.names], np.int32)
        L = lib()
        c, cp = _i32(self.card)
        self.h = C.c_void_p(L.ve_net_create(len(self.names), cp))
        for node, cpt in spec["cpts"].items():
            vars_ = [self.id[n] for n in cpt["names"]]
            codes = [[self.dom[n].index(lab) for n, lab in zip(cpt["names"], r[:-1])]
                     for r in cpt["rows"]]
            vals = [float(r[-1]) for r in cpt["rows"]]
            v, vp = _i32(vars_)
            k, kp = _i32(np.array(codes, np.int32).reshape(-1))
            x, xp = _f64(vals)
            rc = L.ve_net_set_cpt(self.h, self.id[node], len(vars_), vp, len(vals), kp, xp)
            assert rc == 0, rc

    def __del__(self):
        try:
            lib().ve_net_destroy(self.h)
        except Exception:
            pass

    def query_codes(self, qvars, evars, ecodes, order=None):
        """ids/codes in -> (codes[n, nq] int32, values[n])."""
        L = lib()
        cap = int(np.prod([int(self.card[q]) for q in qvars]))
        q, qp = _i32(qvars)
        e, ep = _i32(evars if len(evars) else [0])
        c, cp = _i32(ecodes if len(ecodes) else [0])
        oc = np.zeros((cap, len(qvars)), np.int32)
        ov = np.zeros(cap, np.float64)
        op = None
        if order is not None:
            o, op = _i32(order)
        n = L.ve_query(self.h, len(qvars), qp, len(evars), ep, cp, op, cap,
                       oc.ctypes.data_as(C.POINTER(C.c_int32)),
                       ov.ctypes.data_as(C.POINTER(C.c_double)))
        if n < 0:
            raise RuntimeError(f"ve_query failed: {n}")
        return oc[:n], ov[:n]

    def last_stats(self):
        a, b = C.c_double(), C.c_double()
        lib().ve_last_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def query(self, query, event, order=None):
        """names/labels in -> (sorted query names, [label tuples], values) as BayesNet.query()
        orders them (levels sorted by name, rows sorted; bayes_net.py:872-875)."""
        qs = sorted(query)
        ev = list(event.items()) if isinstance(event, dict) else list(event)
        qv = [self.id[n] for n in qs]
        evars = [self.id[n] for n, _ in ev]
        ecodes = [_code_of(self.dom[n], lab) for n, lab in ev]
        # out-of-domain evidence (-1) cannot be stored as uint8: use an unused code
        ecodes = [c if c >= 0 else 255 for c in ecodes]
        codes, vals = self.query_codes(qv, evars, ecodes, order)
        labels = [tuple(self.dom[n][int(k)] for n, k in zip(qs, row)) for row in codes]
        return qs, labels, vals
