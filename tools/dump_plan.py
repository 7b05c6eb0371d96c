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
    """Decode a step program (planner.h encoding: 10-word header, GENERIC / FIBER bodies)."""
    steps = []
    n = int(w[0]); p = 1
    for _ in range(n):
        w0 = int(w[p]); kind = w0 & 0xff; n_in = (w0 >> 8) & 0xff; na = (w0 >> 16) & 0xff; nlo = (w0 >> 24) & 0xff
        cx = int(w[p + 1]) & 0xffff; fin = (int(w[p + 1]) >> 16) & 1
        lo, hi, words = int(w[p + 2]), int(w[p + 3]), int(w[p + 6])
        d = dict(w1=int(w[p + 1]), kind={0: "GENERIC", 1: "FIBER", 2: "SWEEP"}[kind], n_in=n_in, na=na, nlo=nlo, fin=fin, cx=cx, lo=lo, hi=hi,
                 bytes=32 * int(w[p + 9]), words=words)
        q = p + 10
        I = lambda k: int(np.int32(w[k]))
        if kind == 2:  # SWEEP: k variables per pass, tile in LDS (planner.h)
            k = na
            d.update(k=k, rb=nlo, kout=int(w[p + 7]) & 0xffff, T=int(w[p + 7]) >> 16, stages=[])
            sq = q + 2 + 5 * k
            for j in range(k):
                s0, s1 = int(w[q + 2 + 5 * j]), int(w[q + 3 + 5 * j])
                ns, nc = (s0 >> 8) & 15, (s0 >> 12) & 15
                d["stages"].append(dict(dig=s0 & 15, cout=(s0 >> 4) & 15, ns=ns, loop=(s0 >> 16) & 15,
                                        fields=[(s0 >> 20) & 15, (s0 >> 24) & 15, (s0 >> 28) & 15],
                                        ctrl=[int(w[q + 4 + 5 * j + c]) & 0xff for c in range(nc)], t_cells=s1 >> 16))
                sq += 7 * ns
        elif kind == 0:
            d["ins"] = [("C" if int(w[q + 3 * j + 1]) >> 31 else "A", I(q + 3 * j + 2)) for j in range(n_in)]
            q += 3 * n_in
            d["card"] = [int(x) for x in w[q:q + na]]; q += na
            d["strides"] = [[I(q + j * na + a) for a in range(na)] for j in range(n_in)]
        else:
            w7 = int(w[p + 7]); nb, ns, nN, nctrl, NC = w7 & 0xf, (w7 >> 4) & 0xf, (w7 >> 8) & 0xf, (w7 >> 12) & 0xf, w7 >> 16
            nT = nN + nctrl
            d.update(nb=nb, ns=ns, nN=nN, nctrl=nctrl, NC=NC, T=int(w[p + 8]) & 0xffff, c1=int(w[p + 8]) >> 16,
                     contig=(int(w[p + 1]) >> 17) & 1)
            d["big"] = [("C" if int(w[q + 4 * b + 1]) >> 31 else "A", (I(q + 4 * b + 2), I(q + 4 * b + 3))) for b in range(nb)]; q += 4 * nb
            d["small"] = [("C" if int(w[q + k * (4 + nT) + 1]) >> 31 else "A", (I(q + k * (4 + nT) + 2), I(q + k * (4 + nT) + 3)),
                           [I(q + k * (4 + nT) + 4 + t) for t in range(nT)]) for k in range(ns)]; q += ns * (4 + nT)
            d["tcard"] = [int(x) for x in w[q:q + nT]]; q += nT
            d["nout"] = [int(x) for x in w[q:q + NC]]; q += NC
            d["outer"] = (int(w[p + 1]) >> 18) & 1
            d["chain"] = (int(w[p + 1]) >> 19) & 1
            if d["outer"]:
                d["nB"] = [int(x) for x in w[q:q + NC]]; q += NC
            if d["chain"]:  # big record 1 = (T12 cells, T3 cells, x3 stride in F, x3 stride in T12); bst[1] = T3 strides
                n3s, nd3 = int(w[q]), int(w[q + 1]) & 0xff
                d["chain3"] = dict(T12=int(w[p + 14]), T3=int(w[p + 15]), fx3=I(p + 16), t12x3=I(p + 17), n12dep=(int(w[q + 1]) >> 8) & 1,
                                   tcard3=[int(x) for x in w[q + 2:q + 2 + nd3]],
                                   small3=[("C" if int(w[q + 2 + nd3 + k * (2 + nd3) + 1]) >> 31 else "A",
                                            [I(q + 2 + nd3 + k * (2 + nd3) + 2 + t) for t in range(nd3)]) for k in range(n3s)])
                q += 2 + nd3 + n3s * (2 + nd3)
            d["rax"] = [(int(w[q + 3 * a]), I(q + 3 * a + 1), I(q + 3 * a + 2)) for a in range(na)]; q += 3 * na
            d["bst"] = [[I(q + b * na + a) for a in range(na)] for b in range(nb)]
        steps.append(d)
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
        if s["lo"] * s["hi"] < int(os.environ.get("MINCELLS", "1")):
            continue
        head = f"{k:3d} {s['kind']:7s} cx={s['cx']} lo={s['lo']} hi={s['hi']} nlo={s['nlo']} na={s['na']} MB={s['bytes']/1e6:8.3f}"
        if s["kind"] == "SWEEP":
            print(head, f"k={s['k']} kout={s['kout']} T={s['T']} stages=" + " ".join(f"[d{g['dig']} out{g['cout']} ns{g['ns']} ctrl{g['ctrl']} loop{g['loop']}]" for g in s["stages"]))
        elif s["kind"] == "GENERIC":
            print(head, f"n_in={s['n_in']} card={s['card']} ins={s['ins']} strides={s['strides']}", "FINAL" if s["fin"] else "")
        else:
            if s.get("chain"):
                head += f" CHAIN {s['chain3']}"
            print(head, f"nb={s['nb']} ns={s['ns']} c1={s['c1']} NC={s['NC']} contig={s['contig']} outer={s['outer']} rs={(s['w1'] >> 20) & 0xff} nctrl={s['nctrl']} T={s['T']} big={s['big']} small={[(t, x) for t, x, _ in s['small']]} "
                        f"tcard={s['tcard']} nout={s['nout']} rax(card,ost,tst)={s['rax']} bst={s['bst']}")
