"""One-off host flattening: pandas CPTs -> dense strided fp64 factor tensors.

Reads what `BayesNet.prepare()` leaves in `bn.P` (sorobn/bayes_net.py:327-371): one float (or int)
Series per node whose (Multi)Index level names are variable names, normally `[*parents, node]`.
Rows may be missing (sparse factors), and - a quirk of the reference, SURVEY.md section 3.4 -
non-root CPTs may be unsorted, so everything is derived from the level *values* of each row, never
from row order.

Output layout (the `mibn_set_network` contract, include/mibn.h): variable ids follow `bn.nodes`
(topological order), each variable's label domain is the sorted union of the labels it takes in
any CPT, and factor v is a dense C-order table over its scope with 0.0 for absent rows.  This is
pure integer index math and must be bit-exact; tests/test_host_logic.py::test_flatten_is_bit_exact checks it
against the sparse rows.
"""
import numpy as np
import pandas as pd


class FlatNetwork:
    __slots__ = ("names", "id", "domains", "dom_index", "card", "scope", "scope_off",
                 "scope_vars", "value_off", "values", "present", "hints", "missing", "parents", "_lut")

    def code_of(self, var_id, label):
        """Evidence labels match by Python equality (`get_level_values(var) == val`,
        bayes_net.py:774), so 1 matches True; -1 when the label is outside the domain."""
        # hashable labels: one dict lookup (equal labels hash equally - 1, 1.0 and True share a slot); anything the
        # dict cannot answer falls through to the equality scan
        try:
            lut = self._lut
        except AttributeError:
            lut = self._lut = {}
        table = lut.get(var_id)
        if table is None:
            table = {}
            try:
                for i, d in enumerate(self.domains[var_id]):
                    table.setdefault(d, i)
            except TypeError:
                table = False
            lut[var_id] = table
        if table is not False:
            try:
                hit = table.get(label)
                if hit is not None:
                    return hit
            except TypeError:
                pass
        for i, d in enumerate(self.domains[var_id]):
            try:
                if d == label:
                    return i
            except Exception:  # pragma: no cover - exotic label types
                pass
        return -1


def _level_names(P, node):
    names = list(P.index.names)
    if len(names) == 1 and names[0] is None:
        names = [node]
    return names


def flatten(bn) -> FlatNetwork:
    """Flatten any object with the reference's `nodes` / `parents` / `P` attributes."""
    names = list(bn.nodes)
    for extra in bn.P:  # BayesNet() with CPTs but no structure (test_bayes_net.py:116-120)
        if extra not in names:
            names.append(extra)
    # level names may be str subclasses / equal-but-distinct objects: key on equality
    ident = {}
    for i, n in enumerate(names):
        ident[n] = i

    scopes, level_values, tables, missing = {}, {}, {}, set()
    labels = [set() for _ in names]
    for v, node in enumerate(names):
        if node not in bn.P:
            missing.add(v)
            continue
        P = bn.P[node]
        lv_names = _level_names(P, node)
        try:
            scope = [ident[n] for n in lv_names]
        except KeyError as e:
            raise KeyError(e.args[0])
        if v not in scope:
            raise ValueError(f"CPT of {node!r} has no level named {node!r} (levels: {lv_names})")
        cols = [np.asarray(P.index.get_level_values(k)) for k in range(len(scope))]
        # canonical scope order: node last (the reference's [*parents, node]); other levels keep
        # their order.  Only an internal layout choice: results never depend on it.
        k_self = scope.index(v)
        order = [k for k in range(len(scope)) if k != k_self] + [k_self]
        scopes[v] = [scope[k] for k in order]
        level_values[v] = [cols[k] for k in order]
        tables[v] = P.to_numpy(dtype=np.float64)
        for u, col in zip(scopes[v], level_values[v]):
            labels[u].update(col.tolist())

    domains, dom_index = [], []
    for v, node in enumerate(names):
        try:
            dom = sorted(labels[v])
        except TypeError as e:
            raise TypeError(f"labels of {node!r} are not mutually comparable: {e}")
        domains.append(dom)
        dom_index.append(pd.Index(dom, name=node) if dom else pd.Index([], name=node))

    card = np.array([max(1, len(d)) for d in domains], np.int32)
    scope_off, scope_vars, value_off, chunks, pchunks = [0], [], [0], [], []
    for v in range(len(names)):
        if v in missing:
            sc = [v]
            dense = np.ones(int(card[v]), np.float64)  # never read: queries touching it raise KeyError
            pres = dense
        else:
            sc = scopes[v]
            shape = [int(card[u]) for u in sc]
            dense = np.zeros(int(np.prod(shape, dtype=np.int64)), np.float64)
            pres = np.zeros(len(dense), np.float64)  # 1.0 where the sparse CPT has a row (even with p = 0)
            if len(tables[v]):
                flat = np.zeros(len(tables[v]), np.int64)
                for u, col in zip(sc, level_values[v]):
                    codes = pd.Index(domains[u]).get_indexer(pd.Index(col))
                    if (codes < 0).any():  # mixed bool/int labels: fall back to equality search
                        lut = domains[u]
                        codes = np.array([next(i for i, d in enumerate(lut) if d == x)
                                          for x in col.tolist()], np.int64)
                    flat = flat * int(card[u]) + codes
                dense[flat] = tables[v]
                pres[flat] = 1.0
        scope_vars += sc
        scope_off.append(len(scope_vars))
        chunks.append(dense)
        pchunks.append(pres)
        value_off.append(value_off[-1] + len(dense))

    fn = FlatNetwork()
    fn.names = names
    fn.id = ident
    fn.domains = domains
    fn.dom_index = dom_index
    fn.card = card
    fn.scope = [scope_vars[a:b] for a, b in zip(scope_off[:-1], scope_off[1:])]
    fn.scope_off = np.array(scope_off, np.int64)
    fn.scope_vars = np.array(scope_vars, np.int32)
    fn.value_off = np.array(value_off, np.int64)
    fn.values = np.concatenate(chunks) if chunks else np.zeros(0, np.float64)
    fn.present = np.concatenate(pchunks) if pchunks else np.zeros(0, np.float64)
    fn.missing = missing
    fn.parents = [[u for u in sc if u != v] for v, sc in enumerate(fn.scope)]
    # elimination-order hint: rank by sorted name (row-major on zero-padded grid ids; the order the
    # hash-ordered reference uses, oracle/refload.py)
    try:
        rank = np.empty(len(names), np.int32)
        rank[np.array(sorted(range(len(names)), key=lambda i: names[i]), np.int64)] = \
            np.arange(len(names), dtype=np.int32)
        fn.hints = [rank]
    except TypeError:
        fn.hints = []
    return fn
