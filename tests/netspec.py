"""Serializable Bayesian-network specs shared by the golden-vector generator and the tests.

A *spec* is plain JSON data:

    {"name": str,
     "hashed_names": bool,          # reference-side only: wrap names in oracle.refload.HashedName
     "nodes": [name, ...],          # every node (isolated ones included)
     "edges": [[parent, child], ...],
     "cpts": {node: {"names": [level names], "rows": [[label, ..., p], ...]}}}

`build(spec, cls)` instantiates it with any class exposing the reference's constructor /
`P` / `prepare()` interface (bayes_net.py:286-371) - the unmodified reference in the build
container, `sorobn_amd.BayesNet` everywhere.  Row order inside a CPT is preserved on purpose: the
reference leaves non-root CPTs unsorted after `prepare()` and the host
flattening must not depend on row order.

The generators follow SURVEY.md Appendix A (grid recipe) and section 8c (random DAGs).
"""
import itertools
import json

import numpy as np
import pandas as pd


# ----------------------------------------------------------------------------- build / dump

def build(spec, cls, wrap=None):
    """Instantiate `spec` with BayesNet class `cls`; `wrap` maps a name str to the name object."""
    w = (lambda s: s) if wrap is None else wrap
    edges = [(w(p), w(c)) for p, c in spec["edges"]]
    in_edges = {n for e in spec["edges"] for n in e}
    isolated = [w(n) for n in spec["nodes"] if n not in in_edges]
    bn = cls(*edges, *isolated)
    for node, cpt in spec["cpts"].items():
        names = [w(n) for n in cpt["names"]]
        rows = cpt["rows"]
        p = [r[-1] for r in rows]
        if len(names) == 1:
            idx = pd.Index([r[0] for r in rows], name=names[0])
        else:
            idx = pd.MultiIndex.from_tuples([tuple(r[:-1]) for r in rows], names=names)
        bn.P[w(node)] = pd.Series(p, index=idx)
    bn.prepare()
    return bn


def _py(v):
    """numpy scalar -> JSON-able python scalar (bool stays bool, int stays int)."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return float(v)
    return str(v)


def dump(bn, name, hashed_names=False):
    """Serialize a prepared reference-style BayesNet (reads bn.P / bn.parents / bn.nodes)."""
    cpts = {}
    for node, P in bn.P.items():
        names = [str(n) for n in P.index.names]
        rows = []
        for key, val in zip(P.index.tolist(), P.tolist()):
            key = key if isinstance(key, tuple) else (key,)
            rows.append([_py(k) for k in key] + [_py(val)])
        cpts[str(node)] = {"names": names, "rows": rows}
    edges = [[str(p), str(c)] for c, ps in bn.parents.items() for p in ps]
    return {"name": name, "hashed_names": hashed_names, "nodes": [str(n) for n in bn.nodes],
            "edges": edges, "cpts": cpts}


# ----------------------------------------------------------------------------- generators

def grid_spec(R, C, K, seed=0, name=None):
    """R x C grid, parents = top & left neighbour, K states 0..K-1, Dirichlet(1) CPT rows drawn in
    topological (anti-diagonal, ties lexicographic) node order from default_rng(seed).

    Same recipe as SURVEY.md Appendix A: node id r*C+c, zero-padded 3-digit name, parents sorted,
    parent configurations in C-order with the child fastest.
    """
    import graphlib

    nm = lambda r, c: f"{r * C + c:03d}"
    parents = {}
    edges = []
    for r in range(R):
        for c in range(C):
            ps = []
            if r:
                ps.append(nm(r - 1, c))
            if c:
                ps.append(nm(r, c - 1))
            parents[nm(r, c)] = sorted(ps)
            edges += [[p, nm(r, c)] for p in sorted(ps)]
    ts = graphlib.TopologicalSorter()
    for n in sorted(parents):
        ts.add(n, *parents[n])
    order = list(ts.static_order())  # == reference bn.nodes (bayes_net.py:319-322)
    rng = np.random.default_rng(seed)
    cpts = {}
    for node in order:
        pa = parents[node]
        cols = [*pa, node]
        p = rng.dirichlet(np.ones(K), size=K ** len(pa)).reshape(-1)
        rows = [list(cfg) + [float(v)] for cfg, v in
                zip(itertools.product(range(K), repeat=len(cols)), p)]
        cpts[node] = {"names": cols, "rows": rows}
    return {"name": name or f"grid{R}x{C}k{K}s{seed}", "hashed_names": True,
            "nodes": order, "edges": edges, "cpts": cpts}


def mixed_grid_spec(R, C, cards, seed):
    """Grid like netspec.grid_spec with a cardinality per COLUMN (cards[c])."""
    import graphlib
    nm = lambda r, c: f"{r * C + c:03d}"
    parents, edges, card_of = {}, [], {}
    for r in range(R):
        for c in range(C):
            ps = ([nm(r - 1, c)] if r else []) + ([nm(r, c - 1)] if c else [])
            parents[nm(r, c)] = sorted(ps)
            card_of[nm(r, c)] = cards[c]
            edges += [[p, nm(r, c)] for p in sorted(ps)]
    ts = graphlib.TopologicalSorter()
    for n in sorted(parents):
        ts.add(n, *parents[n])
    order = list(ts.static_order())
    rng = np.random.default_rng(seed)
    cpts = {}
    for node in order:
        cols = [*parents[node], node]
        doms = [range(card_of[n]) for n in cols]
        n_cfg = int(np.prod([card_of[n] for n in parents[node]])) if parents[node] else 1
        p = rng.dirichlet(np.ones(card_of[node]), size=n_cfg).reshape(-1)
        cpts[node] = {"names": cols, "rows": [list(cfg) + [float(v)] for cfg, v in zip(itertools.product(*doms), p)]}
    return {"name": f"mixed{R}x{C}s{seed}", "hashed_names": True, "nodes": order, "edges": edges, "cpts": cpts}


def random_dag_spec(seed, n_nodes=None, max_parents=3, cards=(2, 3, 4, 5), p_zero=0.08,
                    p_missing=0.05, labels="int", max_cells=None):
    """Random DAG with mixed cardinalities, Dirichlet CPTs, some exact zeros and missing rows.  `max_cells` (cardinalities in the
    hundreds): parents are dropped from the end until a CPT has at most that many cells (consumes no random numbers)."""
    rng = np.random.default_rng(seed)
    n = int(n_nodes or rng.integers(4, 13))
    names = [f"{i:03d}" for i in range(n)]
    card = [int(rng.choice(cards)) for _ in range(n)]
    if labels == "str":
        dom = [[f"s{j}" for j in range(card[i])] for i in range(n)]
    else:
        dom = [list(range(card[i])) for i in range(n)]
    edges, parents = [], {}
    for i in range(n):
        k = int(rng.integers(0, min(i, max_parents) + 1))
        ps = sorted(rng.choice(i, size=k, replace=False).tolist()) if k else []
        while max_cells and ps and card[i] * int(np.prod([card[p] for p in ps])) > max_cells:
            ps = ps[:-1]
        parents[i] = ps
        edges += [[names[p], names[i]] for p in ps]
    cpts = {}
    for i in range(n):
        ps = parents[i]
        rows = []
        for cfg in itertools.product(*[dom[p] for p in ps]):
            p = rng.dirichlet(np.ones(card[i]))
            zero = rng.random(card[i]) < p_zero
            if zero.all():
                zero[0] = False
            p = np.where(zero, 0.0, p)
            p = p / p.sum()
            for j, lab in enumerate(dom[i]):
                if ps and rng.random() < p_missing:
                    continue  # missing row: the reference's factors are sparse
                rows.append([*cfg, lab, float(p[j])])
        # a label that never appears in a node's own CPT would shrink its domain: keep one row
        seen = {r[-2] for r in rows}
        for lab in dom[i]:
            if lab not in seen:
                cfg = [dom[p][0] for p in ps]
                rows.append([*cfg, lab, 0.0])
        # the reference needs unique index rows
        uniq = {}
        for r in rows:
            uniq[tuple(r[:-1])] = r
        cpts[names[i]] = {"names": [names[p] for p in ps] + [names[i]], "rows": list(uniq.values())}
    return {"name": f"dag{seed}", "hashed_names": True, "nodes": names, "edges": edges, "cpts": cpts}


def domains(spec):
    """node -> sorted list of labels (union over every CPT that mentions the node)."""
    dom = {n: set() for n in spec["nodes"]}
    for cpt in spec["cpts"].values():
        for r in cpt["rows"]:
            for nme, lab in zip(cpt["names"], r[:-1]):
                dom[nme].add(lab)
    return {n: sorted(v) for n, v in dom.items()}


# ----------------------------------------------------------------------------- request streams

def c3_requests(n_nodes, K, n, n_evidence=4, seed=1, start=0):
    """The BASELINE C3 stream: per request 1 query node + n_evidence evidence
    nodes uniform without replacement, evidence states uniform in 0..K-1, from default_rng(seed).
    Returns int32 arrays (qvar[n], evars[n, n_evidence], ecodes[n, n_evidence]); requests
    start..start+n-1 of the stream (the stream is generated sequentially, so a prefix is stable)."""
    rng = np.random.default_rng(seed)
    total = start + n
    q = np.empty(total, np.int32)
    ev = np.empty((total, n_evidence), np.int32)
    ec = np.empty((total, n_evidence), np.int32)
    # drawn in blocks so the stream does not depend on `n`
    B = 4096
    for lo in range(0, total, B):
        hi = min(total, lo + B)
        m = B
        perm = np.argsort(rng.random((m, n_nodes)), axis=1)[:, :n_evidence + 1].astype(np.int32)
        codes = rng.integers(0, K, size=(m, n_evidence)).astype(np.int32)
        q[lo:hi] = perm[:hi - lo, 0]
        ev[lo:hi] = perm[:hi - lo, 1:]
        ec[lo:hi] = codes[:hi - lo]
    return q[start:], ev[start:], ec[start:]


def grid_row_major_cost(q, evs, R, C, K):
    """Scope-only cost of eliminating a grid request in ascending-id (row-major) order, the order the
    hash-ordered reference uses: returns (total product rows, largest product rows)."""
    par = lambda v: ([v - C] if v >= C else []) + ([v - 1] if v % C else [])
    evs = set(int(e) for e in evs)
    rel, stack = set(), [int(q), *evs]
    while stack:
        v = stack.pop()
        if v not in rel:
            rel.add(v)
            stack += par(v)
    fs = [frozenset(u for u in par(v) + [v] if u not in evs) for v in rel]
    total = biggest = 0
    for x in sorted(rel - {int(q)} - evs):
        ins = [s for s in fs if x in s]
        fs = [s for s in fs if x not in s]
        u = frozenset().union(*ins)
        total += K ** len(u)
        biggest = max(biggest, K ** len(u))
        fs.append(u - {x})
    u = frozenset().union(*fs)
    return total + K ** len(u), max(biggest, K ** len(u))


def asia_requests(node_names, n, seed=0):
    """The BASELINE C2 stream: query var uniform over the nodes, n_evidence uniform in {1,2,3} from
    the others, values uniform {False, True}.  Returns a list of (query name, {name: bool})."""
    rng = np.random.default_rng(seed)
    out = []
    nn = len(node_names)
    for _ in range(n):
        perm = rng.permutation(nn)
        ne = int(rng.integers(1, 4))
        vals = rng.integers(0, 2, size=ne)
        out.append((node_names[perm[0]],
                    {node_names[perm[1 + j]]: bool(vals[j]) for j in range(ne)}))
    return out


def save(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))


def load(path):
    with open(path) as f:
        return json.load(f)
