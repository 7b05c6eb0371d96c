"""Parameter and structure learning on grouped counts.

The reference learns CPTs with `X.groupby([*parents, node]).size()` per node (sorobn/bayes_net.py:467-516) and a
Chow-Liu tree from the pairwise `X.groupby([u, v]).size()` (sorobn/structure.py:9-63).  Here the label columns are
factorised once on the host (sorted label domains -> uint8 codes) and *all* contingency tables of a call are counted
by one launch of the count kernel (csrc/count_kernel.hip.h, `mibn_count_tables`); the small tables that come back are
turned into the same pandas objects on the host.  No CPU fallback: counting needs a gfx950 device.
"""
import itertools

import numpy as np
import pandas as pd

from . import _capi

_engines = {}


def counting_engine(device=None):
    """One engine per device, used only for `count_tables` (no network attached)."""
    from .bayes_net import _default_device
    d = _default_device() if device is None else device
    if d not in _engines:
        _engines[d] = _capi.Engine(d)
    return _engines[d]


def encode_columns(X, columns):
    """Label columns -> (uint8 codes [n_rows, n_cols], sorted label domains, cards).  Missing values (NaN / None / NaT)
    are NOT labels: pandas' `groupby(...).size()` and `value_counts()` drop them (dropna=True), so they get the extra
    code len(domain) - `cards[j]` is then len(domain) + 1 - and `count_tables_dropna` cuts that slice off every table.
    More than 256 codes per column is outside what the count kernel (and any CPT one would learn) handles."""
    codes = np.empty((len(columns), len(X)), np.uint8)  # filled column by column: column-major, as the count kernel reads it
    domains, cards = [], []
    for j, c in enumerate(columns):
        v = X[c].to_numpy()
        col = dom = None
        has_na = False
        if v.dtype.kind in "iub" and len(v):  # small integer range: a lookup table instead of hashing (no NA possible)
            iv = v.view(np.uint8) if v.dtype.kind == "b" else v
            lo, hi = int(iv.min()), int(iv.max())
            if hi - lo < (1 << 16):
                shifted = iv.astype(np.int64) - lo  # (in int64: narrow signed dtypes would wrap, e.g. int8 100 - (-100))
                present = np.zeros(hi - lo + 1, bool)
                present[shifted] = True
                dom = (np.flatnonzero(present) + lo).astype(v.dtype)
                if len(dom) <= 256:
                    col = (np.cumsum(present) - 1).astype(np.uint8)[shifted]
        if col is None:
            col, dom = pd.factorize(v, sort=True, use_na_sentinel=True)  # one hash pass per column; NA -> -1
            has_na = bool((col < 0).any())
            if has_na:
                col = np.where(col < 0, len(dom), col)
        if len(dom) + int(has_na) > 256:
            raise ValueError(f"column {c!r} has {len(dom)} distinct labels (max 256)")
        codes[j] = col
        domains.append(pd.Index(dom, name=c))
        cards.append(len(dom) + int(has_na))
    return codes.T, domains, cards


def count_tables_dropna(engine, codes, domains, cards, tables):
    """Dense contingency tables over the *labels*: rows with a missing value in any of a table's columns are dropped
    (the NA slice, where a column has one, is cut off) - pandas' dropna=True."""
    dense = engine.count_tables(codes, cards, tables)
    out = []
    for d, t in zip(dense, tables):
        cut = tuple(slice(0, len(domains[c])) for c in t)
        out.append(np.ascontiguousarray(d[cut]) if any(cards[c] != len(domains[c]) for c in t) else d)
    return out


def count_series(counts, domains, names):
    """Dense contingency table -> the Series `groupby(names).size()` returns: one row per observed combination."""
    flat = counts.reshape(-1)
    keep = np.flatnonzero(flat)
    if len(names) == 1:
        return pd.Series(flat[keep], index=domains[0][keep].rename(names[0]))
    codes = np.unravel_index(keep, counts.shape)
    idx = pd.MultiIndex(levels=[d.rename(n) for d, n in zip(domains, names)], codes=list(codes), names=list(names),
                        verify_integrity=False)
    return pd.Series(flat[keep], index=idx)


def grouped_counts(X, tables, device=None):
    """`tables`: list of column-name tuples -> list of `X.groupby(list(t)).size()`-like Series, one GPU launch."""
    columns = sorted({c for t in tables for c in t}, key=list(X.columns).index)
    codes, domains, cards = encode_columns(X, columns)
    pos = {c: j for j, c in enumerate(columns)}
    dense = count_tables_dropna(counting_engine(device), codes, domains, cards, [tuple(pos[c] for c in t) for t in tables])
    return [count_series(d, [domains[pos[c]] for c in t], list(t)) for d, t in zip(dense, tables)]


# ------------------------------------------------------------------------------------------------ rank 3: fit
def partial_fit(bn, X):
    """BayesNet.partial_fit (bayes_net.py:467-510): update every CPT with the rows of X.  Exact integer counts are
    kept per node (`bn._counts`), so fitting in chunks gives bit-identical CPTs to fitting at once."""
    if not hasattr(bn, "_counts") or bn._counts is None:
        bn._counts = {}
    tables = [tuple([*bn.parents[c], c]) for c in bn.parents] + [(r,) for r in bn.roots]
    fresh = grouped_counts(X, tables, device=getattr(bn, "_device", None))
    for names, new in zip(tables, fresh):
        node = names[-1]
        new = new.astype(np.float64)
        if node in bn._counts:
            counts = bn._counts[node].add(new, fill_value=0.0)
        else:
            counts = new
            if bn.prior_count and len(names) > 1:
                # the reference adds ONE pseudo-count for every combination of the values seen in this chunk, whatever
                # prior_count is (bayes_net.py:480-488)
                combos = pd.MultiIndex.from_tuples(list(itertools.product(*[X[v].unique() for v in names])), names=list(names))
                counts = counts.add(pd.Series(1.0, combos), fill_value=0.0)
        bn._counts[node] = counts
        if len(names) > 1:
            bn.P[node] = counts / counts.groupby(level=list(names[:-1])).transform("sum")
        else:
            # Deliberate deviation for a ROOT column with missing values (NaN / None): the reference divides the counts of
            # the observed labels by the number of ROWS seen so far (`_P_sizes[root] += len(X)`, bayes_net.py:503-508), so
            # its P(root) sums to less than 1 when rows were missing; here the CPT stays a distribution (counts / their
            # sum).  Without missing values in the column the two agree exactly.
            bn.P[node] = counts / counts.sum()
    bn.prepare()
    return bn


def fit(bn, X):
    """BayesNet.fit (bayes_net.py:512-516)."""
    bn.P = {}
    bn._counts = {}
    return partial_fit(bn, X)


# ------------------------------------------------------------------------------------------- rank 4: Chow-Liu
def mutual_information(X, device=None):
    """All pairwise mutual informations (structure.py:33-45, 55-63) from one counting launch: {(u, v): mi} for u < v."""
    cols = sorted(X.columns)
    codes, domains, cards = encode_columns(X, cols)
    pairs = list(itertools.combinations(range(len(cols)), 2))
    dense = count_tables_dropna(counting_engine(device), codes, domains, cards, [(j,) for j in range(len(cols))] + pairs)
    n = float(len(X))
    # value_counts(normalize=True) divides by the non-missing rows of the column, groupby().size() / len(X) by all rows
    # (structure.py:33-41)
    marg = [d / float(d.sum()) if d.sum() else d.astype(np.float64) for d in dense[:len(cols)]]
    out = {}

    def one(i, j, c):
        puv = c / n
        nz = puv > 0
        ratio = puv[nz] / (marg[j][None, :].repeat(len(marg[i]), 0)[nz] * marg[i][:, None].repeat(len(marg[j]), 1)[nz])
        return float((puv[nz] * np.log(ratio)).sum())

    # pairs whose table has no empty cell (the usual case with many rows) are evaluated together, shape by shape -
    # the same elementwise arithmetic and the same row sums as `one`, bit for bit; the others go through `one`
    by_shape = {}
    for k, (i, j) in enumerate(pairs):
        by_shape.setdefault(dense[len(cols) + k].shape, []).append(k)
    mi = np.empty(len(pairs), np.float64)
    for shape, ks in by_shape.items():
        c = np.stack([dense[len(cols) + k] for k in ks])
        full = (c > 0).reshape(len(ks), -1).all(axis=1)
        if full.any():
            sel = np.flatnonzero(full)
            pi = np.stack([marg[pairs[ks[t]][0]] for t in sel])[:, :, None]
            pj = np.stack([marg[pairs[ks[t]][1]] for t in sel])[:, None, :]
            puv = c[sel] / n
            mi[np.asarray(ks)[sel]] = (puv * np.log(puv / (pj * pi))).reshape(len(sel), -1).sum(axis=1)
        for t in np.flatnonzero(~full):
            k = ks[t]
            mi[k] = one(pairs[k][0], pairs[k][1], dense[len(cols) + k])
    for k, (i, j) in enumerate(pairs):
        out[(cols[i], cols[j])] = float(mi[k])
    return out


def chow_liu(X, root=None, device=None):
    """structure.chow_liu (structure.py:9-52): maximum spanning tree of the mutual-information graph (Kruskal with a
    union-find over the edges in descending MI order, ties in sorted-pair order like the reference's stable sort),
    oriented away from `root` (default: the first column).  Returns (parent, child) tuples.  Like the reference's
    `kruskal` (structure.py:108-117) the scan stops as soon as every vertex has a neighbour - which can be before the
    components are joined: the result is then the part of that forest reachable from `root`.
    The ORDER of the returned edge list is unspecified (the reference orients the tree by iterating Python sets, the order
    depends on the hash seed): compare trees as sets of directed edges.
    """
    mi = mutual_information(X, device=device)
    ranked = sorted(mi, key=lambda e: mi[e], reverse=True)  # stable: equal MI keeps combinations() order
    leader = {v: v for v in X.columns}

    def find(v):
        while leader[v] != v:
            leader[v] = leader[leader[v]]
            v = leader[v]
        return v

    size = {v: 1 for v in X.columns}
    adj = {v: [] for v in X.columns}
    touched = 0  # vertices with at least one neighbour
    for u, v in ranked:
        a, b = find(u), find(v)
        if a != b:
            touched += (not adj[u]) + (not adj[v])
            adj[u].append(v)
            adj[v].append(u)
            if size[a] < size[b]:
                a, b = b, a
            leader[b] = a
            size[a] += size[b]
        if touched == len(X.columns):  # structure.py:116-117
            break
    root = X.columns[0] if root is None else root
    edges, stack, seen = [], [root], {root}
    while stack:
        u = stack.pop()
        for v in adj[u]:
            if v not in seen:
                seen.add(v)
                edges.append((u, v))
                stack.append(v)
    return edges
