"""Print the step program the planner emits for one request (debugging aid, CPU only)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import netspec  # noqa: E402
import simengine  # noqa: E402
import sorobn_amd  # noqa: E402
from sorobn_amd.flatten import flatten  # noqa: E402


def program(f, q, ev, codes):
    L = simengine.lib()
    L.plan_sim_program.restype = C.c_int64
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    hints = np.ascontiguousarray(np.stack(f.hints).reshape(-1), np.int32)
    q = np.ascontiguousarray(q, np.int32); ev = np.ascontiguousarray(ev, np.int32); codes = np.ascontiguousarray(codes, np.int32)
    out = np.zeros(1 << 20, np.uint32)
    n = L.plan_sim_program(C.c_int32(len(f.card)), p(f.card, C.c_int32), p(f.scope_off, C.c_int64), p(f.scope_vars, C.c_int32),
                           p(f.value_off, C.c_int64), p(f.values, C.c_double), C.c_int32(len(f.hints)), p(hints, C.c_int32),
                           C.c_int32(len(q)), p(q, C.c_int32), C.c_int32(len(ev)), p(ev, C.c_int32), p(codes, C.c_int32),
                           p(out, C.c_uint32), C.c_int64(len(out)))
    assert n > 0, n
    return out[:n]


def decode(w):
    steps = []
    n = int(w[0]); p = 1
    for _ in range(n):
        w0 = int(w[p]); n_in = w0 & 0xff; na = (w0 >> 8) & 0xff; nlo = (w0 >> 16) & 0xff; fin = (w0 >> 24) & 1
        cx, lo, hi = int(w[p + 1]), int(w[p + 2]), int(w[p + 3]); words = int(w[p + 6])
        ins = []
        for j in range(n_in):
            off = int(w[p + 8 + 3 * j]) | (int(w[p + 9 + 3 * j]) << 32)
            ins.append(("C" if off >> 63 else "A", off & ~(1 << 63), int(np.int32(w[p + 10 + 3 * j]))))
        card = [int(x) for x in w[p + 8 + 3 * n_in:p + 8 + 3 * n_in + na]]
        st = np.array(w[p + 8 + 3 * n_in + na:p + 8 + 3 * n_in + na + n_in * na]).astype(np.int32).reshape(n_in, na) if na else np.zeros((n_in, 0))
        steps.append(dict(n_in=n_in, na=na, nlo=nlo, fin=fin, cx=cx, lo=lo, hi=hi, ins=ins, card=card, strides=st.tolist()))
        p += words
    return steps


if __name__ == "__main__":
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    f = flatten(bn)
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    qs, evs, ecs = netspec.c3_requests(100, 4, 4096, 4, seed=1)
    i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    steps = decode(program(f, [to_var[qs[i]]], to_var[evs[i]], ecs[i]))
    print("request", i, "q", qs[i], "ev", evs[i], "steps", len(steps))
    for k, s in enumerate(steps):
        if s["lo"] * s["hi"] >= int(os.environ.get("MINCELLS", "1")):
            print(k, f"n_in={s['n_in']} cx={s['cx']} cells={s['lo']*s['hi']:8d} lo={s['lo']} hi={s['hi']} nlo={s['nlo']} card={s['card']}",
                  " | ".join(f"{t}{'' if t=='A' else ''} xs={xs} s={st}" for (t, o, xs), st in zip(s["ins"], s["strides"])))
