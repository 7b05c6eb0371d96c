"""Host-side mirror of the reference's `BayesNet` for the exact-inference hot path.

Same constructor, attributes (`P`, `parents`, `children`, `nodes`), `prepare()`, `query()` and
`impute()` as MaxHalford/sorobn (sorobn/bayes_net.py:259-371, 796-908) - names, argument meaning,
return types and error behaviour - but `query(algorithm="exact")` runs on an MI355X: the pandas
CPTs are flattened once (flatten.py) and the variable-elimination loop (bayes_net.py:739-794) is
executed by hand-written gfx950 kernels behind the C-ABI of include/mibn.h.  There is no CPU
fallback: without the HIP extension or without a gfx950 device `query` raises.

`full_joint_dist` / `predict_proba` / `predict_log_proba` (bayes_net.py:398-465, 934-973; SURVEY.md section 8f rank
1) ride on the same kernels: the joint is the posterior of all variables given no evidence, the likelihood of
partially observed rows is the posterior of the observed columns given no evidence (variable elimination of the
rest - so it also works where the reference's full joint would not fit).

`sample`, `query(algorithm="rejection" / "likelihood")` (bayes_net.py:518-663; section 8f rank 2) run on a forward-
sampling kernel (one sample per lane); like Gibbs their parity is statistical (the reference's stream needs `vose`).

`fit` / `partial_fit` (467-516; rank 3) and `sorobn_amd.structure.chow_liu` (structure.py:9-52; rank 4) count their
contingency tables with one launch of the count kernel (learning.py).

`sorobn_amd.examples` offers the reference's four example networks; `is_tree`, `markov_boundary`, `iter_dfs` mirror the
small graph helpers.  Out of scope: graph drawing (`graphviz`), GUI, CLI.

`accelerate(bn)` attaches the same backend to an *existing reference object* by replacing the two
methods `query` dispatches to (bayes_net.py:848, 851-853); see INTEGRATION.md.
"""
import collections
import collections.abc
import graphlib
import itertools
import operator
import os
import sys
import types

import numpy as np
import pandas as pd

from . import _capi
from .flatten import flatten

__all__ = ["BayesNet", "accelerate", "Backend"]


def _default_device():
    for key in ("MIBN_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    return 0


class _Ragged(Exception):
    """exact_many: a request whose arity differs from the first one's (the pipeline takes fixed-arity sub-batches)."""


_get_index = operator.attrgetter("index")
_get_values = operator.attrgetter("_values")  # the Series' own ndarray, no copy (`.values` builds a view object per call)
_get_names = operator.attrgetter("names")
_get_pnames = operator.attrgetter("_names")
_get_pname = operator.attrgetter("_name")


class PosteriorBatch(collections.abc.Sequence):
    """The answers of `BayesNet.query_many`: a read-only sequence whose item i is exactly the Series `query(*q_i, event=e_i)`
    returns (built when asked for: a pandas Series costs ~14 us to construct, a batch of 2^18 of them four seconds - more than
    the kernels), over the dense posteriors as they came back from the device.
      `batch[i]`, `len`, iteration, slices -> Series / lists of Series
      `batch.dense(i)`   the dense C-order posterior over request i's query variables (caller order)
      `batch.to_frame()` every answer in ONE pandas object, built vectorised: a long Series indexed by (request, *labels) when all
                         requests ask for the same variables, else a DataFrame with one row per positive cell"""

    def __init__(self, backend, queries, out, out_off):
        self._b, self._q, self.out, self.out_off = backend, queries, out, out_off

    def __len__(self):
        return len(self._q)

    def dense(self, i):
        return self.out[self.out_off[i]:self.out_off[i + 1]]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._b.finished_series(self._q[i], self.dense(i))

    def to_frame(self):
        """One pandas object for the whole batch.  Same query variables in every request: a Series `p` with a MultiIndex
        (request, *sorted query names) - level values and row order inside a request exactly those of `batch[i]`, so
        `frame.xs(i, level="request")` equals `batch[i]` (up to the name).  Otherwise (the query variable changes from request to
        request): a DataFrame with columns request, variables (tuple of names), cell (C-order cell of the finished answer's index)
        and p, one row per positive cell."""
        n = len(self)
        sizes = np.diff(self.out_off)
        req = np.repeat(np.arange(n, dtype=np.int64), sizes)
        same = n > 0 and all(q == self._q[0] for q in self._q)
        if same:
            name, order, levels, shape, names = self._b._tail(self._q[0])
            out = self.out
            if shape is None:
                keep = np.flatnonzero(out > 0)
                codes = [req[keep], keep - self.out_off[req[keep]]]
                lv = [pd.RangeIndex(n, name="request"), levels]
                nm = ["request", self._q[0][0]]
            else:
                cells = int(np.prod(shape))
                if order is not None:
                    out = np.ascontiguousarray(out.reshape([n] + shape).transpose([0] + [k + 1 for k in order])).reshape(-1)
                    shape = [shape[k] for k in order]
                keep = np.flatnonzero(out > 0)
                codes = [keep // cells, *np.unravel_index(keep % cells, shape)]
                lv = [pd.RangeIndex(n, name="request"), *levels]
                nm = ["request", *names]
            idx = pd.MultiIndex(levels=lv, codes=codes, names=nm, verify_integrity=False)
            return pd.Series(out[keep], index=idx, name="p")
        keep = np.flatnonzero(self.out > 0)
        r = req[keep]
        # (cells of the caller-order dense table; `batch[i]` gives the finished Series of a request)
        variables = np.empty(n, object)
        variables[:] = self._q
        return pd.DataFrame({"request": r, "variables": variables[r], "cell": keep - self.out_off[r], "p": self.out[keep]})


class CptWatch:
    """Has anything the flattened tables were built from changed?  The reference re-reads `P` on every query (bayes_net.py:770),
    so a CPT replaced (`bn.P['A'] = s`), edited in place (`bn.P['A'][True] = 0.9`, also through a reference the caller kept) or
    re-indexed must reach the device tables.  Round 4 re-hashed every CPT on every `query()` (several hundred
    microseconds on the 100-node grid).  Now: identity of every Series, of its index and of its value array (three id lists
    compared in C), the level names, and ONE bitwise comparison of all CPT numbers against a snapshot (`np.concatenate` of the live
    arrays: 44 KB on the 10x10 grid) - 5 us on a five-node network, 60 us on the grid, and exact: no hashing, no version counter a
    held reference could bypass.  CPTs whose values are not a numeric ndarray fall back to the hashed
    fingerprint."""

    def __init__(self, bn):
        items = list(bn.P.items())
        self.keys = [k for k, _ in items]
        self.series = [v for _, v in items]  # (kept alive: their ids below cannot be recycled by other objects)
        self.exact = True
        try:
            self.index = list(map(_get_index, self.series))
            arrays = list(map(_get_values, self.series))
            self.exact = all(isinstance(a, np.ndarray) and a.dtype != object and a.ndim == 1 for a in arrays)
        except AttributeError:
            self.exact = False
        if self.exact:
            self.ids = (list(map(id, self.series)), list(map(id, self.index)))
            # level names live on the index object and can be re-assigned in place; pandas keeps them in `_names` (MultiIndex: a
            # list) / `_name` (Index) - read directly where present (checked against the public accessor here), the public
            # FrozenList-building `.names` costs ten times as much per CPT
            self.multi = [ix for ix in self.index if isinstance(ix, pd.MultiIndex)]
            self.single = [ix for ix in self.index if not isinstance(ix, pd.MultiIndex)]
            try:
                self.private_names = (all(list(ix._names) == list(ix.names) for ix in self.multi) and
                                      all(ix._name == ix.name or (ix._name is None and ix.name is None) for ix in self.single))
            except AttributeError:
                self.private_names = False
            if self.private_names:
                self.names = ([list(ix._names) for ix in self.multi], [ix._name for ix in self.single])
            else:
                self.names = [list(n) for n in map(_get_names, self.index)]
            self.snap = np.concatenate(arrays).tobytes() if arrays else b""
        else:
            self.slow = Backend.fingerprint_of(bn)

    def changed(self, bn) -> bool:
        P = bn.P
        if len(P) != len(self.keys):
            return True
        if not self.exact:
            return Backend.fingerprint_of(bn) != self.slow
        # (every loop below runs inside map / list comparison: no Python-level work per CPT)
        ids_s, ids_i = self.ids
        if list(map(id, map(P.get, self.keys))) != ids_s or list(map(id, map(_get_index, self.series))) != ids_i:
            return True
        if self.private_names:
            if list(map(_get_pnames, self.multi)) != self.names[0] or list(map(_get_pname, self.single)) != self.names[1]:
                return True
        elif list(map(_get_names, self.index)) != self.names:
            return True
        if not self.series:
            return False
        # the LIVE value arrays (a Series may have replaced its array - an upcast, copy-on-write) against the snapshot, bit for bit
        try:
            return np.concatenate(list(map(_get_values, self.series))).tobytes() != self.snap
        except (ValueError, TypeError):
            return True


class Backend:
    """Flattened network + one mibn engine.  Built lazily from any object exposing the reference's
    `nodes` / `parents` / `P`; rebuilt when the CPT objects change."""

    def __init__(self, bn, device=None, planner_only=False):
        self.flat = flatten(bn)
        self.watch = CptWatch(bn)
        self.engine = _capi.Engine(_default_device() if device is None else device,
                                   planner_only=planner_only)
        f = self.flat
        self.engine.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.values)
        if f.hints:
            self.engine.set_order_hints(np.stack(f.hints))
        self._anc = {}
        self._presence = None
        self._device = device
        self._planner_only = planner_only

    def presence_engine(self):
        """A second engine over the same structure whose tables hold 1.0 where the sparse CPT has a row: its joint
        is positive exactly on the rows of the reference's inner join (full_joint_dist(keep_zeros=True))."""
        if self._presence is None:
            f = self.flat
            e = _capi.Engine(_default_device() if self._device is None else self._device, planner_only=self._planner_only)
            e.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.present)
            self._presence = e
        return self._presence

    def marginal(self, names, engine=None):
        """Dense C-order table P(names) with no evidence (all other variables eliminated)."""
        q, _, _ = self.encode(names, {})
        eng = self.engine if engine is None else engine
        # every CPT takes part (bayes_net.py:460): no pruning to the ancestors - with sparse CPTs a barren node
        # does not sum to 1.  A per-call flag: nothing engine-wide changes under asynchronous calls in flight.
        return eng.query_fixed([q], np.zeros((1, 0), np.int32), np.zeros((1, 0), np.int32), flags=_capi.Q_NOPRUNE)[0]

    def joint_series(self, names, keep_zeros=False):
        """Series over the sorted `names` like `full_joint_dist` builds it (bayes_net.py:460-465): sorted levels, sorted
        rows, zero rows dropped unless keep_zeros (then: the rows present in every CPT)."""
        f = self.flat
        ids = [f.id[n] for n in names]
        dense = self.marginal(names)
        if keep_zeros:
            mask = self.marginal(names, self.presence_engine()) > 0
        else:
            mask = dense > 0
        keep = np.flatnonzero(mask)
        if len(ids) == 1:
            idx = f.dom_index[ids[0]][keep].rename(names[0])
        else:
            codes = np.unravel_index(keep, [int(f.card[v]) for v in ids])
            idx = pd.MultiIndex(levels=[f.dom_index[v] for v in ids], codes=list(codes), names=list(names),
                                verify_integrity=False)
        return pd.Series(dense[keep], index=idx)

    def stale(self, bn) -> bool:
        """True when `bn.P` no longer is what this backend was flattened from (`CptWatch`)."""
        return self.watch.changed(bn)

    @staticmethod
    def fingerprint_of(bn):
        """Identity AND content of the CPTs by hashing (the slow path of `CptWatch`: CPTs whose values are not numeric
        ndarrays)."""
        fp = []
        for k, v in bn.P.items():
            vals = getattr(v, "values", None)
            try:
                if vals is None:
                    content = None
                elif vals.dtype != object:
                    content = hash(np.ascontiguousarray(vals).tobytes())
                else:
                    content = hash(tuple(vals.tolist()))
            except (TypeError, ValueError, AttributeError):
                content = None
            idx = getattr(v, "index", None)
            if isinstance(idx, pd.MultiIndex):
                ih = hash(tuple(np.ascontiguousarray(c).tobytes() for c in idx.codes)) ^ hash(tuple(len(l) for l in idx.levels))
            elif idx is not None:
                try:
                    ih = hash(tuple(idx.tolist()))
                except TypeError:
                    ih = len(idx)
            else:
                ih = 0
            fp.append((id(k), id(v), len(v), content, ih))
        return tuple(fp)

    # ---- id / code conversion -------------------------------------------------------------------
    def var_id(self, name):
        try:
            return self.flat.id[name]
        except (KeyError, TypeError):
            raise KeyError(name)  # bayes_net.py:770 / 373-375: unknown node -> KeyError(name)

    def _ancestors(self, v):
        if v not in self._anc:
            s = set()
            for p in self.flat.parents[v]:
                s.add(p)
                s |= self._ancestors(p)
            self._anc[v] = s
        return self._anc[v]

    def encode(self, query, event):
        """names/labels -> (qvars, evars, ecodes); raises KeyError like the reference when a relevant
        node is unknown or has no CPT."""
        q = [self.var_id(n) for n in query]
        ev = [self.var_id(n) for n in event]
        codes = [self.flat.code_of(v, lab) for v, lab in zip(ev, event.values())]
        if self.flat.missing:
            rel = set(q) | set(ev)
            for v in list(rel):
                rel |= self._ancestors(v)
            for v in sorted(rel & self.flat.missing):
                raise KeyError(self.flat.names[v])
        return q, ev, codes

    # ---- result construction --------------------------------------------------------------------
    def posterior_series(self, query, dense):
        """Dense C-order posterior over `query` (caller order) -> the Series
        `_variable_elimination` returns (bayes_net.py:789-794): zero rows absent, one level per query
        variable, plain Index for a single variable."""
        f = self.flat
        ids = [f.id[n] for n in query]
        keep = np.flatnonzero(dense > 0)
        vals = dense[keep]
        if len(ids) == 1:
            idx = f.dom_index[ids[0]][keep]
            idx = idx.rename(query[0])
            return pd.Series(vals, index=idx)
        shape = [int(f.card[v]) for v in ids]
        codes = np.unravel_index(keep, shape)
        idx = pd.MultiIndex(levels=[f.dom_index[v] for v in ids], codes=list(codes),
                            names=list(query), verify_integrity=False)
        return pd.Series(vals, index=idx)

    def _tail(self, query):
        """What the tail of `query` (bayes_net.py:869-875: rename, level sort, row sort) amounts to for a given query tuple, worked
        out once: the Series name, and for several variables the level permutation into sorted-name order.  The dense posterior
        is C-order over `query`, the label domains are sorted, so transposing it into sorted-name order and listing its positive
        cells in C-order IS the sorted index - no pandas reorder_levels / sort_index per answer (61 -> 14 us on this host)."""
        try:
            cache = self._tails
        except AttributeError:
            cache = self._tails = {}
        t = cache.get(query)
        if t is None:
            f = self.flat
            ids = [f.id[n] for n in query]
            name = f"P({', '.join(query)})"
            if len(ids) == 1:
                t = (name, None, f.dom_index[ids[0]].rename(query[0]), None, None)
            else:
                names = list(query)
                order = [names.index(n) for n in sorted(names)]
                shape = [int(f.card[v]) for v in ids]
                t = (name, order if order != list(range(len(ids))) else None, [f.dom_index[ids[k]] for k in order],
                     shape, [names[k] for k in order])
            if len(cache) < 4096:
                cache[query] = t
        return t

    def finished_series(self, query, dense):
        """Dense C-order posterior over `query` (caller order) -> exactly the Series `query()` returns: zero rows absent
        (bayes_net.py:789-794), named `P(...)`, levels in sorted-name order, rows sorted (869-875)."""
        name, order, levels, shape, names = self._tail(query)
        if shape is None:
            keep = np.flatnonzero(dense > 0)
            if len(keep) == len(dense):
                # (a view of the cached Index - renaming the answer's index must not reach the cache - and, for an answer that is a
                # slice of a batch's buffer, its own numbers: ADVICE r5)
                return pd.Series(dense if dense.base is None else dense.copy(), index=levels.view(), name=name)
            return pd.Series(dense[keep], index=levels[keep], name=name)
        if order is not None:
            dense = np.ascontiguousarray(dense.reshape(shape).transpose(order)).reshape(-1)
            shape = [shape[k] for k in order]
        keep = np.flatnonzero(dense > 0)
        idx = pd.MultiIndex(levels=levels, codes=list(np.unravel_index(keep, shape)), names=names, verify_integrity=False)
        return pd.Series(dense[keep], index=idx, name=name)

    def exact_query(self, query, event):
        """`query(*query, event=event)` of the exact path, finished: encode, ONE call into the library, the answer Series."""
        q, ev, codes = self.encode(query, event)
        one = getattr(self.engine, "query_one", None)
        dense = one(q, ev, codes) if one is not None else self.engine.query_fixed([q], [ev], [codes])[0]
        return self.finished_series(query, dense)

    # ---- the two replaced methods ---------------------------------------------------------------
    def variable_elimination(self, *query, event):
        q, ev, codes = self.encode(query, event)
        out = self.engine.query_fixed([q], [ev], [codes])
        return self.posterior_series(query, out[0])

    def variable_elimination_many(self, requests):
        """requests: iterable of (query tuple, event dict) -> list of Series (one launch)."""
        requests = list(requests)
        q_off, e_off, qv, evs, ecs = self.encode_many(requests)
        out, out_off = self.engine.query_batch(q_off, qv, e_off, evs, ecs)
        return [self.posterior_series(query, out[a:b])
                for (query, _), a, b in zip(requests, out_off[:-1], out_off[1:])]

    def encode_many(self, requests):
        """CSR arrays (q_off, q_vars, e_off, e_vars, e_codes) of a list of (query tuple, event dict).  Names and labels go
        through plain dict lookups in bulk (one pass over the flattened names, one over the flattened labels) instead of one
        `encode` call with its set algebra per request; anything the bulk path cannot answer - an unknown name, an unhashable or
        out-of-domain label, a network with missing CPTs - falls back to `encode` request by request, which raises the
        reference's KeyError or codes the label -1 as before."""
        f = self.flat
        n = len(requests)
        try:
            if f.missing:
                raise KeyError
            ident = f.id
            nq = np.fromiter((len(q) for q, _ in requests), np.int64, n)
            ne = np.fromiter((len(e) for _, e in requests), np.int64, n)
            qv = np.fromiter((ident[name] for q, _ in requests for name in q), np.int32, int(nq.sum()))
            evs = np.fromiter((ident[name] for _, e in requests for name in e), np.int32, int(ne.sum()))
            luts = self._label_luts()
            ecs = np.fromiter((luts[v][lab] for v, lab in zip(evs.tolist(), (lab for _, e in requests for lab in e.values()))),
                              np.int32, len(evs))
            q_off = np.zeros(n + 1, np.int64)
            e_off = np.zeros(n + 1, np.int64)
            np.cumsum(nq, out=q_off[1:])
            np.cumsum(ne, out=e_off[1:])
            return q_off, qv, e_off, evs, ecs
        except (KeyError, TypeError):
            pass
        q_off, e_off, qv, evs, ecs = [0], [0], [], [], []
        for query, event in requests:
            q, ev, codes = self.encode(query, event)
            qv += q
            evs += ev
            ecs += codes
            q_off.append(len(qv))
            e_off.append(len(evs))
        return (np.array(q_off, np.int64), np.array(qv, np.int32), np.array(e_off, np.int64), np.array(evs, np.int32),
                np.array(ecs, np.int32))

    def _label_luts(self):
        """Per variable: label -> code (equal labels hash equally: 1, 1.0 and True share a slot, like the `==` of
        bayes_net.py:774); a variable with unhashable labels has no table (KeyError -> the per-request path)."""
        try:
            return self._luts
        except AttributeError:
            luts = []
            for dom in self.flat.domains:
                t = {}
                try:
                    for i, d in enumerate(dom):
                        t.setdefault(d, i)
                except TypeError:
                    t = {}
                luts.append(t)
            self._luts = luts
            return luts

    def exact_many(self, requests, sub_batch=32768):
        """The exact path over a list of (query tuple, event dict) -> `PosteriorBatch`.  Encoded in bulk, sent to the engine in
        sub-batches of `sub_batch` requests with two calls in flight (mibn_submit_batch / mibn_wait: sub-batch k + 1 is encoded
        and planned while the kernels of k run) when the batch is fixed-arity; a ragged batch goes as one blocking call."""
        def checked(chunk):
            """Query tuples normalised (a bare name is a 1-tuple) and the request checks of `query` (bayes_net.py:840-845)."""
            chunk = [((q,) if isinstance(q, str) else tuple(q), e) for q, e in chunk]
            for q, e in chunk:
                if not q:
                    raise ValueError("At least one query variable has to be specified")
                if not e.keys().isdisjoint(q):
                    raise ValueError("A query variable cannot be part of the event")
            return chunk

        n = len(requests)
        eng = self.engine
        if n == 0:
            return PosteriorBatch(self, [], np.zeros(0), np.zeros(1, np.int64))
        if not (hasattr(eng, "submit_fixed") and n > sub_batch):
            requests = checked(requests)
            q_off, qv, e_off, evs, ecs = self.encode_many(requests)
            out, out_off = eng.query_batch(q_off, qv, e_off, evs, ecs)
            return PosteriorBatch(self, [q for q, _ in requests], out, np.asarray(out_off, np.int64))
        # a long batch: sub-batches through the two-deep pipeline.  Checked, normalised and encoded sub-batch by sub-batch on a helper
        # thread (a malformed request in sub-batch k raises after the sub-batches before it have run; nothing is returned); the
        # pipeline takes fixed-arity batches - the arity of the first request - and a ragged one falls back to ONE blocking call
        q00 = requests[0][0]
        nq0, ne0 = (1 if isinstance(q00, str) else len(q00)), len(requests[0][1])
        queries = []
        # names / labels -> ids / codes on a helper thread, one sub-batch ahead: the engine calls below release the GIL (ctypes), so
        # the encoding of sub-batch k + 1 runs while the planner works on k and the host waits for the kernels of k - 1
        import queue
        import threading
        todo = queue.Queue(maxsize=2)
        stop = threading.Event()  # set by the consumer when it gives up: the producer does not encode the rest of a 1 M-request batch
        # query tables of different sizes in one fixed-arity batch (a 2-state and a 3-state query variable): the engine's flat
        # wait; an engine without one (the test doubles) takes such a batch through the general CSR call
        wait_flat = getattr(eng, "wait_flat", None)

        def produce():
            try:
                for a in range(0, n, sub_batch):
                    if stop.is_set():
                        break
                    chunk = checked(requests[a:a + sub_batch])
                    if not all(len(q) == nq0 and len(e) == ne0 for q, e in chunk):
                        raise _Ragged()
                    _, qv, _, evs, ecs = self.encode_many(chunk)
                    cells = np.prod(eng.card[qv.reshape(len(chunk), nq0)].astype(np.int64), axis=1)
                    if wait_flat is None and len(cells) and cells.min() != cells.max():
                        raise _Ragged()
                    queries.extend(q for q, _ in chunk)
                    todo.put((len(chunk), qv, evs, ecs, cells))
            except BaseException as e:  # noqa: BLE001 - re-raised on the calling thread (the reference's KeyError for an unknown name)
                todo.put(e)
            todo.put(None)

        def collect(h):
            return (wait_flat(h) if wait_flat is not None else eng.wait(h)).reshape(-1)

        worker = threading.Thread(target=produce, daemon=True)
        worker.start()
        parts, pending, cells = [], None, []
        try:
            while True:
                item = todo.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                b, qv, evs, ecs, c = item
                h = eng.submit_fixed(qv.reshape(b, nq0), evs.reshape(b, ne0), ecs.reshape(b, ne0))
                if pending is not None:
                    prev, pending = pending, h  # (cleared before the wait: an error inside it must not wait on the same ticket again)
                    parts.append(collect(prev))
                pending = h
                cells.append(c)
        except BaseException:
            stop.set()
            last, pending = pending, None
            if last is not None:  # (a call in flight writes into its pinned result buffer: collect it before the error leaves)
                try:
                    eng.wait(last)
                except Exception:  # noqa: BLE001
                    pass
            while worker.is_alive():  # let the producer finish (its queue may be full)
                try:
                    todo.get(timeout=0.05)
                except queue.Empty:
                    pass
            if isinstance(sys.exc_info()[1], _Ragged):  # requests of different arity: the general CSR call, once
                requests = checked(requests)
                q_off, qv, e_off, evs, ecs = self.encode_many(requests)
                out, out_off = eng.query_batch(q_off, qv, e_off, evs, ecs)
                return PosteriorBatch(self, [q for q, _ in requests], out, np.asarray(out_off, np.int64))
            raise
        worker.join()
        parts.append(collect(pending))
        out_off = np.zeros(n + 1, np.int64)
        np.cumsum(np.concatenate(cells), out=out_off[1:])
        return PosteriorBatch(self, queries, np.concatenate(parts), out_off)

    def gibbs_sampling(self, *query, event, n_iterations, n_chains=1, seed=0):
        q, ev, codes = self.encode(query, event)
        f = self.flat
        # the reference cycles through sorted(nodes - event) (bayes_net.py:697,718)
        free = [v for v in range(len(f.names)) if v not in set(ev)]
        try:
            cycle = sorted(free, key=lambda v: f.names[v])
        except TypeError:
            cycle = free
        counts = self.engine.gibbs(q, ev, codes, n_chains, n_iterations, seed=seed, cycle=cycle)
        dense = counts.astype(np.float64) / float(max(1, n_chains * n_iterations))
        return self.posterior_series(query, dense)

    # ---- forward sampling and the algorithms built on it ---------------------------
    def _cells_series(self, query, values, present):
        """Series over the joint query states `present` (flat C-order cell ids), like a pandas groupby builds it."""
        f = self.flat
        ids = [f.id[n] for n in query]
        keep = np.flatnonzero(present)
        if len(ids) == 1:
            idx = f.dom_index[ids[0]][keep].rename(query[0])
        else:
            codes = np.unravel_index(keep, [int(f.card[v]) for v in ids])
            idx = pd.MultiIndex(levels=[f.dom_index[v] for v in ids], codes=list(codes), names=list(query),
                                verify_integrity=False)
        return pd.Series(np.asarray(values, np.float64)[keep], index=idx)

    def rejection_sampling(self, *query, event, n_iterations, seed=0):
        """bayes_net.py:577-619: the share of each query state among the forward samples that agree with the event."""
        q, ev, codes = self.encode(query, event)
        _, counts = self.engine.sampling_query(1, q, ev, codes, n_iterations, seed=seed)
        total = counts.sum()
        return self._cells_series(query, counts / float(total) if total else counts.astype(np.float64), counts > 0)

    def likelihood_weighting(self, *query, event, n_iterations, seed=0):
        """bayes_net.py:621-663: per query state the mean likelihood of the samples drawn with the event clamped,
        normalised (the reference's estimator, product of P(value | parents) over all nodes)."""
        q, ev, codes = self.encode(query, event)
        if any(c < 0 for c in codes):  # P.get(value, 0) == 0 for a label outside the domain: every likelihood is 0
            raise ValueError("likelihood weighting: an event label is outside the variable's domain")
        wsum, counts = self.engine.sampling_query(2, q, ev, codes, n_iterations, seed=seed)
        mean = np.divide(wsum, counts, out=np.zeros_like(wsum), where=counts > 0)
        s = mean.sum()
        return self._cells_series(query, mean / s if s > 0 else mean, counts > 0)

    def forward_samples(self, n, init, seed=0):
        """bayes_net.py:518-575: n joint samples as a DataFrame of labels (columns in variable-id = `nodes` order)."""
        f = self.flat
        iv = [self.var_id(k) for k in init]
        ic = [f.code_of(v, lab) for v, lab in zip(iv, init.values())]
        if any(c < 0 for c in ic):
            raise ValueError("sample: an init label is outside the variable's domain")
        codes = self.engine.sample(n, iv, ic, seed=seed)
        return pd.DataFrame({name: np.asarray(f.dom_index[v])[codes[:, v]] for v, name in enumerate(f.names)})


class BayesNet:
    """Bayesian network with an MI355X exact-inference backend.

    Parameters follow the reference (bayes_net.py:259-289): `structure` is a sequence of
    `(parent(s), child(ren))` tuples (lists are expanded to their cartesian product) and bare node
    names; `prior_count` and `seed` are accepted for signature compatibility (`seed` seeds the Gibbs
    kernel).
    """

    def __init__(self, *structure, prior_count: int = None, seed: int = None):
        self.prior_count = prior_count
        self.seed = seed
        as_list = lambda o: o if isinstance(o, list) else [o]
        parents = collections.defaultdict(set)
        children = collections.defaultdict(set)
        loose = set()
        for item in structure:
            if isinstance(item, tuple):
                ps, cs = item
                for p, c in itertools.product(as_list(ps), as_list(cs)):
                    parents[c].add(p)
                    children[p].add(c)
            else:
                loose.add(item)
        self.parents = {n: sorted(ps) for n, ps in parents.items()}
        self.children = {n: sorted(cs) for n, cs in children.items()}
        # topological order, ties lexicographic - same construction as bayes_net.py:319-322 so that
        # `nodes` (and with it variable ids) are identical to the reference's
        sorter = graphlib.TopologicalSorter()
        for n in sorted({*self.parents, *self.children, *loose}):
            sorter.add(n, *self.parents.get(n, []))
        self.nodes = list(sorter.static_order())
        self.P = {}
        self._backend = None
        self._device = None

    # ---- pickling / copying: device handles are rebuilt lazily -----------
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_backend"] = None
        return d

    # ---- CPT house-keeping (bayes_net.py:327-371) ------------------------------------------------
    def prepare(self) -> "BayesNet":
        """Normalise the user's CPTs: DataFrame with a 'p' column -> Series; name / order the
        index levels as [*parents, node]; sort; name the Series.  Raises the reference's
        ValueErrors (bayes_net.py:341-344, 349-352)."""
        for node in list(self.P):
            P = self.P[node]
            parents = self.parents.get(node, [])
            if isinstance(P, pd.DataFrame):
                if "p" not in P.columns:
                    raise ValueError(f"DataFrame for '{node}' must have a 'p' column "
                                     f"containing probabilities")
                cols = [c for c in P.columns if c != "p"]
                expected = set(parents) | {node}
                if set(cols) != expected:
                    raise ValueError(f"DataFrame for '{node}' has columns {cols}, "
                                     f"but expected {sorted(expected)} (plus 'p')")
                P = P.set_index([*parents, node])["p"]
            wanted = [*parents, node]
            if not parents:
                P = P.copy() if P is self.P[node] else P
                P.index.name = node
            elif set(P.index.names) == set(wanted):
                # NOTE: the reference computes this reordering but never stores it back
                # (SURVEY.md section 3.4); here the reordered, sorted table is what is kept
                P = P.reorder_levels(wanted)
            else:
                P = P.copy() if P is self.P[node] else P
                P.index.names = wanted
            P = P.sort_index()
            P.name = f"P({node} | {', '.join(map(str, parents))})" if parents else f"P({node})"
            self.P[node] = P
        self._backend = None
        return self

    # ---- graph helpers on the hot path ----------------------------------------------------------
    def ancestors(self, node):
        """Set of ancestors (bayes_net.py:373-378), memoised instead of re-walked."""
        memo = {}

        def walk(n):
            if n not in memo:
                s = set()
                for p in self.parents.get(n, ()):
                    s.add(p)
                    s |= walk(p)
                memo[n] = s
            return memo[n]

        return set(walk(node))

    @property
    def roots(self):
        return [n for n in self.nodes if n not in self.parents]

    @property
    def leaves(self):
        return [n for n in self.nodes if n not in self.children]

    @property
    def is_tree(self):
        """No node has more than one parent (bayes_net.py:975-1000)."""
        return all(len(ps) <= 1 for ps in self.parents.values())

    def markov_boundary(self, node):
        """Parents, children and the children's other parents, sorted (bayes_net.py:1002-1039) - the variables the
        Gibbs kernel reads when it resamples `node`."""
        boundary = set(self.parents.get(node, []))
        for child in self.children.get(node, []):
            boundary.add(child)
            boundary.update(self.parents[child])
        boundary.discard(node)
        return sorted(boundary)

    def iter_dfs(self):
        """Depth-first pre-order from every root in `roots` order, children in sorted order (bayes_net.py:1041-1075)."""
        seen = set()
        stack = list(reversed(self.roots))
        while stack:
            node = stack.pop()
            if node in seen:
                continue
            seen.add(node)
            yield node
            stack.extend(c for c in reversed(self.children.get(node, [])) if c not in seen)

    # ---- backend --------------------------------------------------------------------------------
    def use_device(self, device: int):
        """Bind to a HIP device index (default: $MIBN_DEVICE, $LOCAL_RANK, else 0)."""
        self._device = device
        self._backend = None
        return self

    @property
    def backend(self) -> Backend:
        b = self._backend
        if b is None or b.stale(self):
            b = self._backend = Backend(self, device=self._device)
        return b

    def _variable_elimination(self, *query, event):
        return self.backend.variable_elimination(*query, event=event)

    def _gibbs_sampling(self, *query, event, n_iterations, n_chains=1):
        return self.backend.gibbs_sampling(*query, event=event, n_iterations=n_iterations,
                                           n_chains=n_chains, seed=self.seed or 0)

    def _next_seed(self):
        """The reference advances one `random.Random(seed)` per object (bayes_net.py:289): successive sampling
        calls see different streams, the same sequence of calls on an equally seeded object repeats."""
        self._draws = getattr(self, "_draws", 0) + 1
        return ((self.seed or 0) * 0x9E3779B97F4A7C15 + self._draws) & (2**64 - 1)

    def _rejection_sampling(self, *query, event, n_iterations):
        return self.backend.rejection_sampling(*query, event=event, n_iterations=n_iterations, seed=self._next_seed())

    def _llh_weighting(self, *query, event, n_iterations):
        return self.backend.likelihood_weighting(*query, event=event, n_iterations=n_iterations, seed=self._next_seed())

    def sample(self, n=1, init: dict = None, method="forward"):
        """Forward (ancestral) samples, bayes_net.py:550-575: a Series for n == 1, else a DataFrame with sorted
        columns; `init` forces variables to given values."""
        if method != "forward":
            raise ValueError("Unknown method, must be one of: forward")
        df = self.backend.forward_samples(max(1, n), init or {}, seed=self._next_seed())
        if n > 1:
            return df.sort_index(axis="columns")
        return df.iloc[0].rename(None)

    # ---- public API (bayes_net.py:796-908) ------------------------------------------------------
    @staticmethod
    def _check_request(query, event):
        if not query:
            raise ValueError("At least one query variable has to be specified")
        for q in query:
            if q in event:
                raise ValueError("A query variable cannot be part of the event")

    @staticmethod
    def _finish(answer, query):
        """rename + level sort + row sort, exactly the tail of `query` (bayes_net.py:869-875)."""
        answer = answer.rename(f"P({', '.join(query)})")
        if isinstance(answer.index, pd.MultiIndex):
            answer = answer.reorder_levels(sorted(answer.index.names))
        return answer.sort_index()

    def query(self, *query, event: dict, algorithm="exact", n_iterations=100, n_chains=1) -> pd.Series:
        """Answer a probabilistic query; same contract as the reference (bayes_net.py:796-875).

        `algorithm`: "exact" (variable elimination on the GPU), "gibbs" (GPU chains; `n_chains` is an
        extension, default 1 like the reference), "rejection" or "likelihood" (GPU forward sampling, one sample
        per lane); anything else raises the reference's ValueError.
        """
        self._check_request(query, event)
        if algorithm == "exact":
            # (the hook the reference's query() dispatches to, bayes_net.py:848, is honoured: not re-bound on this object - e.g. by
            # accelerate() - and not overridden by a subclass)
            if "_variable_elimination" not in self.__dict__ and type(self)._variable_elimination is BayesNet._variable_elimination:
                return self.backend.exact_query(query, event)  # the finished Series, built directly (Backend._tail)
            answer = self._variable_elimination(*query, event=event)
        elif algorithm == "gibbs":
            answer = self._gibbs_sampling(*query, event=event, n_iterations=n_iterations,
                                          n_chains=n_chains)
        elif algorithm == "rejection":
            answer = self._rejection_sampling(*query, event=event, n_iterations=n_iterations)
        elif algorithm == "likelihood":
            answer = self._llh_weighting(*query, event=event, n_iterations=n_iterations)
        else:
            raise ValueError("Unknown algorithm, must be one of: exact, gibbs, likelihood, "
                             + "rejection")
        return self._finish(answer, query)

    def query_many(self, requests, sub_batch=32768):
        """Batched extension: `requests` = iterable of (query tuple, event dict) -> `PosteriorBatch`, a sequence whose item i is
        the Series `query(*q_i, event=e_i)` returns (identical: name, index, level order, row order, values).  Names and labels are
        encoded in bulk, the device works through sub-batches with two calls in flight, and the Series are built on access;
        `.to_frame()` gives every answer as one pandas object."""
        return self.backend.exact_many(requests if isinstance(requests, list) else list(requests), sub_batch=sub_batch)

    def query_frame(self, *query, events: pd.DataFrame) -> pd.DataFrame:
        """The fixed-query batch in pandas' own shape (an extension; the reference answers one event per call, bayes_net.py:796):
        row r of `events` is an event - its columns are evidence variables, NaN / None = not observed in that row - and row r of
        the result is the posterior of `query` given that event: columns = the joint states of the query variables in the order of
        `query()`'s index (sorted names, sorted labels; a MultiIndex for several variables), zeros where `query()` drops the row, so
        that `out.iloc[r][out.iloc[r] > 0]` has the values of `query(*query, event=<row r>)`.  Labels are encoded per column
        (`Index.get_indexer`), rows are grouped by their set of observed columns, every group is one fixed-shape engine call."""
        if not query:
            raise ValueError("At least one query variable has to be specified")
        if any(c in query for c in events.columns):
            raise ValueError("A query variable cannot be part of the event")
        be = self.backend
        f = be.flat
        q = [be.var_id(n) for n in query]
        if f.missing:
            be.encode(query, {c: None for c in events.columns})  # (raises the reference's KeyError for a node without a CPT)
        cols = list(events.columns)
        ev_ids = np.array([be.var_id(c) for c in cols], np.int32)
        n = len(events)
        codes = np.empty((n, len(cols)), np.int32)
        observed = np.empty((n, len(cols)), bool)
        for j, c in enumerate(cols):
            col = events[c]
            observed[:, j] = col.notna().to_numpy()
            codes[:, j] = pd.Index(f.domains[ev_ids[j]]).get_indexer(col) if len(f.domains[ev_ids[j]]) else -1
        name, order, levels, shape, names = be._tail(tuple(query))
        cells = int(np.prod([int(f.card[v]) for v in q]))
        out = np.zeros((n, cells), np.float64)
        # one engine call per pattern of observed columns
        pat = observed @ (1 << np.arange(len(cols), dtype=np.int64)) if len(cols) < 63 else None
        groups = ([np.arange(n)] if len(cols) == 0 else
                  [np.flatnonzero(pat == p) for p in np.unique(pat)] if pat is not None else
                  [np.array([r]) for r in range(n)])
        qarr = np.array(q, np.int32)
        for rows in groups:
            if not len(rows):
                continue
            on = np.flatnonzero(observed[rows[0]])
            out[rows] = be.engine.query_fixed(np.broadcast_to(qarr, (len(rows), len(q))), np.broadcast_to(ev_ids[on], (len(rows), len(on))),
                                              codes[np.ix_(rows, on)])
        if shape is None:
            columns = levels
        else:
            if order is not None:
                out = np.ascontiguousarray(out.reshape([n] + shape).transpose([0] + [k + 1 for k in order])).reshape(n, cells)
                shape = [shape[k] for k in order]
            columns = pd.MultiIndex(levels=levels, codes=list(np.unravel_index(np.arange(cells), shape)), names=names, verify_integrity=False)
        return pd.DataFrame(out, index=events.index, columns=columns)

    def impute(self, sample: dict, **query_params) -> pd.Series:
        """Replace the `None` entries of `sample` by the most probable joint assignment
        (bayes_net.py:877-908).  With a single missing variable the reference zips the *scalar*
        `idxmax()` (TypeError for bool labels, character-zip for strings - SURVEY.md section 3.2);
        here that case simply works."""
        missing = [k for k, v in sample.items() if v is None]
        event = {k: v for k, v in sample.items() if v is not None}
        posterior = self.query(*missing, event=event, **query_params)
        best = posterior.idxmax()
        if len(missing) == 1:
            best = (best,)
        for k, v in zip(posterior.index.names, best):
            event[k] = v
        return pd.Series(event)

    # ---- the joint and likelihoods (bayes_net.py:398-465, 934-973) -----------------
    def _all_names(self):
        names = []
        for P in self.P.values():
            for n in (P.index.names if P.index.names[0] is not None else [P.name]):
                if n not in names:
                    names.append(n)
        return sorted(names)

    def full_joint_dist(self, event: dict = None, keep_zeros=False) -> pd.Series:
        """The normalised product of all CPTs (bayes_net.py:460-465; like the reference, `event` is accepted and
        ignored).  On the device it is the posterior of all variables given no evidence."""
        names = self._all_names()
        fjd = self.backend.joint_series(names, keep_zeros=keep_zeros)
        fjd.name = f"P({', '.join(names)})"
        return fjd

    def predict_proba(self, X):
        """Likelihood of one sample (dict -> float) or of the rows of a DataFrame (bayes_net.py:934-962).  The
        reference marginalises its full joint onto the observed columns; here the unobserved variables are
        eliminated on the device, which is the same table.  As in the reference, a single observed column returns
        that column's marginal (not indexed by the rows), and a row of probability zero raises KeyError."""
        if isinstance(X, dict):
            return self.predict_proba(pd.DataFrame([X])).iloc[0]
        names = self._all_names()
        observed = [n for n in names if n in set(X.columns)]
        if len(observed) > 1:
            fast = self._predict_proba_rows(X, observed, f"P({', '.join(names)})")
            if fast is not None:
                return fast
        fjd = self.backend.joint_series(observed)
        fjd.name = f"P({', '.join(names)})"
        if len(observed) > 1:
            return fjd[pd.MultiIndex.from_frame(X[observed])]
        return fjd

    def _predict_proba_rows(self, X, observed, name):
        """`fjd[pd.MultiIndex.from_frame(X[observed])]` without the pandas look-up: the rows' labels are encoded column by column
        (`Index.get_indexer` on the sorted label domains), the dense joint is gathered at the raveled codes, and the result carries
        the index pandas' look-up returns - `fjd.index.take(positions)`: the joint's own levels with the rows' codes.  A row that
        is not in the joint (a label outside a domain, probability zero) raises KeyError like the look-up does.  Returns None where
        the columns cannot be encoded in bulk (unhashable / mixed labels): the caller takes the pandas path."""
        be = self.backend
        f = be.flat
        try:
            ids = [f.id[n] for n in observed]
            codes = []
            for n, v in zip(observed, ids):
                c = f.dom_index[v].get_indexer(X[n])
                codes.append(c)
        except (KeyError, TypeError, ValueError, pd.errors.InvalidIndexError):
            return None
        shape = [int(f.card[v]) for v in ids]
        bad = np.zeros(len(X), bool)
        for c in codes:
            bad |= c < 0
        dense = be.marginal(observed)
        flat = np.ravel_multi_index([np.where(bad, 0, c) for c in codes], shape) if len(X) else np.zeros(0, np.int64)
        vals = dense[flat]
        bad |= ~(vals > 0)
        if bad.any():
            rows = X[observed][bad].head(5).itertuples(index=False, name=None)
            raise KeyError(f"{[tuple(r) for r in rows]} not in index")
        idx = pd.MultiIndex(levels=[f.dom_index[v] for v in ids], codes=codes, names=list(observed), verify_integrity=False)
        return pd.Series(vals, index=idx, name=name)

    def predict_log_proba(self, X):
        """bayes_net.py:964-973."""
        return np.log(self.predict_proba(X))

    # ---- parameter learning (bayes_net.py:467-516) ----------------------------------
    def partial_fit(self, X: pd.DataFrame):
        from . import learning
        return learning.partial_fit(self, X)

    def fit(self, X: pd.DataFrame):
        from . import learning
        return learning.fit(self, X)


def accelerate(bn, device=None, backend_factory=None):
    """Attach the MI355X backend to an existing *reference* `sorobn.BayesNet` instance.

    Replaces the two bound methods `BayesNet.query` dispatches to - `_variable_elimination`
    (bayes_net.py:848) and `_gibbs_sampling` (851-853) - and leaves `query`'s own post-processing
    (869-875) and `impute` (877-908) untouched, so naming / sorting stay byte-identical; also replaces
    `full_joint_dist` (398-465), on which the reference's `predict_proba` / `predict_log_proba` build.

    `backend_factory(bn) -> Backend` is a test seam (the GPU-less build container injects the CPU plan simulator of
    tests/simengine.py); the default builds the HIP backend and raises without a gfx950 device.
    """
    state = {"backend": None}
    make = backend_factory or (lambda b: Backend(b, device=device))

    def backend():
        b = state["backend"]
        if b is None or b.stale(bn):
            b = state["backend"] = make(bn)
        return b

    def _variable_elimination(self, *query, event):
        return backend().variable_elimination(*query, event=event)

    def _gibbs_sampling(self, *query, event, n_iterations):
        return backend().gibbs_sampling(*query, event=event, n_iterations=n_iterations,
                                        seed=getattr(self, "seed", None) or 0)

    def full_joint_dist(self, event: dict = None, keep_zeros=False):  # bayes_net.py:398-465
        names = sorted({n for P in self.P.values() for n in P.index.names})
        fjd = backend().joint_series(names, keep_zeros=keep_zeros)
        fjd.name = f"P({', '.join(names)})"
        return fjd

    bn._variable_elimination = types.MethodType(_variable_elimination, bn)
    bn._gibbs_sampling = types.MethodType(_gibbs_sampling, bn)
    bn.full_joint_dist = types.MethodType(full_joint_dist, bn)  # predict_proba / predict_log_proba build on it (952)
    bn._mibn_backend = backend
    return bn
