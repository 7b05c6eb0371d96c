"""CPU tests of the host side: flattening (bit-exact index math), the C++ planner (executed through
the CPU plan simulator, tests/simengine.py), the BayesNet API / error behaviour, and the C-ABI
library (loads, exports every symbol of include/mibn.h, refuses to compute without a device)."""
import copy
import ctypes
import os
import subprocess
import pickle
import re

import numpy as np
import pandas as pd
import pytest

import golden_util as gu
import netspec
import simengine
import sorobn_amd
from sorobn_amd import _capi
from sorobn_amd.flatten import flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nets(fname):
    return gu.load(fname)


# ------------------------------------------------------------------------------------ C-ABI

def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mibn.h")).read()
    declared = set(re.findall(r"\b(mibn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    L = _capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.mibn_version()


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.MibnError) as e:
        _capi.Engine(0)
    assert e.value.code == _capi.E_NODEVICE
    spec = _nets("examples.json")[0]["spec"]
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    with pytest.raises(_capi.MibnError):
        bn.query(spec["nodes"][0], event={})
    # a planner-only context can plan but refuses to answer
    eng = _capi.Engine(planner_only=True)
    f = flatten(bn)
    eng.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.values)
    assert eng.plan_stats([0], [1])["n_steps"] >= 1
    with pytest.raises(_capi.MibnError) as e:
        eng.query_fixed([[0]], [[1]], [[0]])
    assert e.value.code == _capi.E_NODEVICE


def test_set_network_validation():
    eng = _capi.Engine(planner_only=True)
    with pytest.raises(_capi.MibnError):  # scope must end with the variable itself
        eng.set_network([2, 2], [0, 1, 3], [0, 1, 0], [0, 2, 6], np.ones(6))
    with pytest.raises(_capi.MibnError):  # wrong table size
        eng.set_network([2, 2], [0, 1, 3], [0, 0, 1], [0, 2, 5], np.ones(5))
    with pytest.raises(_capi.MibnError):  # cycle
        eng.set_network([2, 2], [0, 2, 4], [1, 0, 0, 1], [0, 4, 8], np.ones(8))
    eng.set_network([2, 2], [0, 1, 3], [0, 0, 1], [0, 2, 6], np.ones(6))
    with pytest.raises(_capi.MibnError):  # query var in the event
        eng.plan_stats([0], [0])
    with pytest.raises(_capi.MibnError):  # unknown id
        eng.plan_stats([7], [])


# ------------------------------------------------------------------------------------ flatten

@pytest.mark.parametrize("fname", ["examples.json", "random_dags.json", "wide_cards.json", "many_nodes.json"])
def test_flatten_is_bit_exact(fname):
    for net in _nets(fname):
        spec = net["spec"]
        bn = netspec.build(spec, sorobn_amd.BayesNet)
        f = flatten(bn)
        dom = netspec.domains(spec)
        assert f.names == list(bn.nodes) + [n for n in bn.P if n not in bn.nodes]
        for v, name in enumerate(f.names):
            assert f.domains[v] == dom[name]
            cpt = spec["cpts"][name]
            sc = f.scope[v]
            assert sc[-1] == v and sorted(f.names[u] for u in sc) == sorted(cpt["names"])
            table = f.values[f.value_off[v]:f.value_off[v + 1]].copy()
            for row in cpt["rows"]:
                lab = dict(zip(cpt["names"], row[:-1]))
                idx = 0
                for u in sc:
                    idx = idx * int(f.card[u]) + dom[f.names[u]].index(lab[f.names[u]])
                assert table[idx] == float(row[-1])  # bit-exact
                table[idx] = 0.0
            assert not table.any()  # absent rows are 0.0


def test_flatten_ignores_row_order_and_level_order():
    spec = copy.deepcopy(next(n for n in _nets("examples.json") if n["spec"]["name"] == "asia")["spec"])
    a = flatten(netspec.build(spec, sorobn_amd.BayesNet))
    rng = np.random.default_rng(0)
    for cpt in spec["cpts"].values():
        rng.shuffle(cpt["rows"])
    b = flatten(netspec.build(spec, sorobn_amd.BayesNet))
    assert np.array_equal(a.values, b.values) and np.array_equal(a.scope_vars, b.scope_vars)
    # a CPT given with permuted level names (test_bayes_net.py:158-194) lands in the same table
    ref = sorobn_amd.BayesNet(("A", "C"), ("B", "C"))
    ref.P["A"] = pd.Series({True: 0.7, False: 0.3})
    ref.P["B"] = pd.Series({True: 0.4, False: 0.6})
    PC = pd.DataFrame({"B": [True, True, True, True, False, False, False, False],
                       "A": [True, True, False, False, True, True, False, False],
                       "C": [True, False, True, False, True, False, True, False],
                       "p": [1, 0, 0, 1, 0.5, 0.5, 0.001, 0.999]})
    ref.P["C"] = PC.set_index(["B", "A", "C"])["p"]
    ref.prepare()
    assert list(ref.P["C"].index.names) == ["A", "B", "C"]
    f = flatten(ref)
    c = f.id["C"]
    t = f.values[f.value_off[c]:f.value_off[c + 1]].reshape(2, 2, 2)  # [A, B, C]
    assert t[1, 0].tolist() == [0.5, 0.5] and t[1, 1].tolist() == [0.0, 1.0] and t[0, 0].tolist() == [0.999, 0.001]


# ------------------------------------------------------------------------------------ planner (simulated)

def _check_requests(bn, requests, ctx, limit=None):
    worst = 0.0
    for r in requests[:limit]:
        name, inames, rows, vals, multi = gu.expected(r)
        ans = bn.query(*r["query"], event={k: v for k, v in r["event"]})
        c = f"{ctx} {r['query']} {r['event']}"
        assert ans.name == name, c
        assert list(ans.index.names) == inames, c
        assert isinstance(ans.index, pd.MultiIndex) == multi, c
        assert ans.dtype == np.float64, c
        gu.assert_rows_equal(ans.index.tolist(), rows, ctx=c)
        if len(vals):
            worst = max(worst, float(np.max(np.abs(ans.to_numpy() - vals))))
    assert worst <= gu.TOL, (ctx, worst)


# small_cells: inputs above this size are "big" -> lowering it forces the FIBER step form (normally only
# used for > 8 KiB tables) onto the small golden networks, mixed cardinalities and sparse CPTs included
# tiling = (big_iters, tile_h): lowering them turns small steps into tiled levels of the level-synchronous
# schedule (normally only steps with >= 4096 lane-iterations, in tiles sized for 512 KiB of traffic)
# fuse: joint elimination of two consecutive variables in one FIBER step (on by default)
@pytest.mark.parametrize("small_cells,tiling,fuse", [(1024, (4096, 0), 1), (1, (2, 1), 1), (6, (8, 3), 1), (1, (2, 1), 0)])
@pytest.mark.parametrize("fname", ["examples.json", "random_dags.json", "wide_cards.json", "many_nodes.json"])
def test_planner_programs_reproduce_reference(fname, small_cells, tiling, fuse):
    for net in _nets(fname):
        bn = simengine.attach(netspec.build(net["spec"], sorobn_amd.BayesNet), small_cells, tiling, fuse)
        _check_requests(bn, net["requests"], net["spec"]["name"], limit=None if small_cells == 1024 else 60)


@pytest.mark.parametrize("small_cells,tiling,fuse", [(1024, (4096, 0), 1), (1, (2, 1), 1), (6, (8, 3), 1), (1, (2, 1), 0)])
def test_planner_programs_reproduce_reference_huge_cardinalities(small_cells, tiling, fuse):
    """Axes of 17 ... 100 states (huge_cards.json): the planner's programs on the CPU plan simulator against the reference."""
    for entry in _nets("huge_cards.json"):
        spec = gu.dag_spec_from_recipe(entry)
        bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet), small_cells, tiling, fuse)
        _check_requests(bn, entry["requests"], spec["name"])


@pytest.mark.parametrize("small_cells,tiling,fuse", [(1024, (4096, 0), 1), (3, (4, 1), 1), (20, (64, 2), 1), (3, (4, 1), 0)])
def test_planner_programs_reproduce_reference_grids(small_cells, tiling, fuse):
    for entry in _nets("grids_small.json"):
        spec = gu.grid_spec_from_recipe(entry)
        bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet), small_cells, tiling, fuse)
        _check_requests(bn, entry["requests"], spec["name"])


def test_planner_programs_reproduce_reference_grid10x10():
    """The BASELINE C3 network, incl. the reference's 403 s worst case P(099 | 000=0) (SURVEY.md
    Appendix A: 0.28215567090257165, ...)."""
    entry = gu.load("grid10x10.json")
    spec = gu.grid_spec_from_recipe(entry)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    _check_requests(bn, entry["requests"], spec["name"])
    fused_bytes = bn.backend.engine.last_stats[0]
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet), fuse=0)  # one variable per pass
    _check_requests(bn, entry["requests"][-4:], spec["name"])
    assert fused_bytes < bn.backend.engine.last_stats[0]  # joint elimination moves fewer bytes
    worst = next(r for r in entry["requests"] if r["query"] == ["099"] and r["event"] == [["000", 0]])
    assert gu.expected(worst)[3].tolist() == [0.28215567090257165, 0.26459023658731734, 0.2448596401736336,
                                              0.20839445233647746]


def test_query_many_matches_query():
    net = next(n for n in _nets("examples.json") if n["spec"]["name"] == "grades")
    bn = simengine.attach(netspec.build(net["spec"], sorobn_amd.BayesNet))
    reqs = [(tuple(r["query"]), {k: v for k, v in r["event"]}) for r in net["requests"][:40]]
    many = bn.query_many(reqs)
    for (q, e), a in zip(reqs, many):
        pd.testing.assert_series_equal(a, bn.query(*q, event=e))


def test_order_choice_never_worse_than_row_major():
    """The executed plan's section-8(d) bytes must not exceed the row-major order's (one of the
    candidates) and is far below it on average for the C3 mix."""
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    f = flatten(bn)
    eng = _capi.Engine(planner_only=True)
    eng.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.values)
    eng.set_order_hints(np.stack(f.hints))
    q, ev, ec = netspec.c3_requests(100, 4, 64, 4, seed=1)
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    ours = np.array([eng.plan_stats([to_var[q[i]]], to_var[ev[i]])["alg_bytes"] for i in range(64)])

    def row_major_bytes(qi, evs):
        par = lambda v: ([v - 10] if v >= 10 else []) + ([v - 1] if v % 10 else [])
        rel, stack = set(), [qi, *evs]
        while stack:
            v = stack.pop()
            if v not in rel:
                rel.add(v)
                stack += par(v)
        fs = [frozenset(u for u in par(v) + [v] if u not in evs) for v in rel]
        total = 0
        for x in sorted(rel - {qi} - set(evs)):
            ins = [s for s in fs if x in s]
            fs = [s for s in fs if x not in s]
            u = frozenset().union(*ins)
            total += 8 * (sum(4 ** len(s) for s in ins) + 4 ** (len(u) - 1))
            fs.append(u - {x})
        u = frozenset().union(*fs)
        return total + 8 * (sum(4 ** len(s) for s in fs) + 4 ** len(u))

    rm = np.array([row_major_bytes(int(q[i]), set(ev[i].tolist())) for i in range(64)])
    assert (ours <= rm * (1 + 1e-12)).all()
    assert ours.mean() < 0.5 * rm.mean()


def test_sweep_form_on_the_simulator():
    """SWEEP steps (up to five variables per pass with the tile resident in LDS, option sweep=5, the default): the
    planner's programs, executed by the simulator the way ve_sweep_kernel executes them (tiles, in-place stages, the lane /
    loop split of the fibers), reproduce the reference's answers on the 10x10 grid, agree with the CHAIN / pair programs
    to the last bits and move at least a fifth fewer bytes."""
    entry = gu.load("grid10x10.json")
    spec = gu.grid_spec_from_recipe(entry)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    assert bn.backend.engine.sweep == 5
    _check_requests(bn, entry["requests"], spec["name"] + " sweep")
    f = flatten(netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet))
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    q, ev, ec = netspec.c3_requests(100, 4, 48, 4, seed=1)
    b = {}
    for name, k in (("chain", 0), ("sweep3", 3), ("sweep4", 4), ("sweep5", 5)):
        eng = simengine.SimEngine(f)
        eng.set_option("sweep", k)
        tot, res = 0.0, []
        for i in range(48):
            res.append(eng._one([to_var[q[i]]], to_var[ev[i]], ec[i]))
            tot += eng.last_stats[0]
        b[name] = (tot, res)
    for name in ("sweep3", "sweep4", "sweep5"):
        for x, y in zip(b["chain"][1], b[name][1]):
            assert float(np.max(np.abs(x - y))) <= 1e-14
    assert b["sweep5"][0] < 0.8 * b["chain"][0]
    assert b["sweep5"][0] <= b["sweep4"][0] <= b["sweep3"][0] <= b["chain"][0]


def test_order_effort_1_answers_the_same_with_fewer_bytes():
    """Round 6, `order_effort` 1 (csrc/order_search.h, planner.cpp::plan_request_rec; the engine's default): more candidate orders and,
    where the byte model's best order is expensive, its best TWO both emitted and the cheaper program kept.  Any elimination order gives
    the reference's posterior (bayes_net.py:779 eliminates in set-iteration order): the programs of effort 1 - with the second emission
    forced for every request, and at the default threshold - executed on the CPU answer like those of effort 0 within 1e-12, on the
    6 x 6 and 10 x 10 four-state grids, and never move more bytes; on the C3 stream fewer."""
    L = simengine.lib()
    try:
        for rows, cols, n_req, n_ev in ((6, 6, 40, 3), (10, 10, 24, 4), (10, 10, 12, 8)):
            spec = netspec.grid_spec(rows, cols, 4, seed=0)
            f = flatten(netspec.build(spec, sorobn_amd.BayesNet))
            n = rows * cols
            to_var = np.array([f.id[f"{i:03d}"] for i in range(n)], np.int32)
            q, ev, ec = netspec.c3_requests(n, 4, n_req, n_ev, seed=3)
            eng = simengine.SimEngine(f)
            got, moved = {}, {}
            for key in ((0, 1e7), (1, 0.0), (1, 1e7)):
                L.plan_sim_set_order_effort(ctypes.c_int(key[0]), ctypes.c_double(key[1]))
                got[key], moved[key] = [], 0.0
                for i in range(n_req):
                    got[key].append(eng._one([to_var[q[i]]], to_var[ev[i]], ec[i]))
                    moved[key] += eng.last_stats[0]
            for key in ((1, 0.0), (1, 1e7)):
                worst = max(float(np.max(np.abs(a - b))) for a, b in zip(got[0, 1e7], got[key]))
                assert worst <= 1e-12, (rows, cols, key, worst)
            # (the second emission can only help; the extra candidates are ranked by the MODEL and may lose a little on a small sample)
            assert moved[1, 0.0] <= moved[1, 1e7] * (1 + 1e-12), (rows, cols, moved)
            if rows == 10:
                assert moved[1, 0.0] < 0.99 * moved[0, 1e7], moved
    finally:
        L.plan_sim_set_order_effort(ctypes.c_int(0), ctypes.c_double(1e7))


def test_chain_form_on_the_simulator():
    """CHAIN steps (three variables per pass, option chain=1): the planner's programs, executed by the simulator,
    reproduce the reference's answers on the 10x10 grid and move fewer bytes than the two-variable passes."""
    entry = gu.load("grid10x10.json")
    spec = gu.grid_spec_from_recipe(entry)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    assert bn.backend.engine.chain == 1  # the default, as in the product
    _check_requests(bn, entry["requests"], spec["name"] + " chain")
    bn.backend.engine.set_option("chain", 0)
    _check_requests(bn, entry["requests"][-6:], spec["name"] + " pairs only")
    f = flatten(netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet))
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    q, ev, ec = netspec.c3_requests(100, 4, 24, 4, seed=1)
    plain, chain = simengine.SimEngine(f), simengine.SimEngine(f)
    plain.set_option("chain", 0)
    plain.set_option("sweep", 0)
    chain.set_option("sweep", 0)
    b0 = b1 = 0.0
    for i in range(24):
        a = plain._one([to_var[q[i]]], to_var[ev[i]], ec[i])
        b0 += plain.last_stats[0]
        b = chain._one([to_var[q[i]]], to_var[ev[i]], ec[i])
        b1 += chain.last_stats[0]
        assert float(np.max(np.abs(a - b))) <= 1e-14
    assert b1 < 0.9 * b0
    tiled = simengine.SimEngine(f, tiling=(64, 3))  # several iterations per tile, CHAIN steps on small tables
    tiled.set_option("sweep", 0)
    for i in range(8):
        assert float(np.max(np.abs(tiled._one([to_var[q[i]]], to_var[ev[i]], ec[i]) - plain._one([to_var[q[i]]], to_var[ev[i]], ec[i])))) <= 1e-14


def test_c3_bayes_rule_and_marginalisation_on_the_simulator():
    """CPU twin of the GPU test of the same name: P(q | e1..e4) against the renormalised slice of P(q, e4 | e1..e3)
    and the marginal of that joint against P(q | e1..e3), planner programs executed by the simulator."""
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    f = flatten(netspec.build(spec, sorobn_amd.BayesNet))
    eng = simengine.SimEngine(f)
    B = 16
    q, ev, ec = netspec.c3_requests(100, 4, B, 4, seed=1)
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    Q, E = to_var[q], to_var[ev]
    cond = eng.query_fixed(Q[:, None], E, ec)
    joint = eng.query_fixed(np.stack([Q, E[:, 3]], 1), E[:, :3], ec[:, :3]).reshape(B, 4, 4)
    prior = eng.query_fixed(Q[:, None], E[:, :3], ec[:, :3])
    assert float(np.max(np.abs(joint.sum(2) - prior))) <= 1e-12
    sl = joint[np.arange(B), :, ec[:, 3]]
    assert float(np.max(np.abs(sl / sl.sum(1, keepdims=True) - cond))) <= 1e-11


def _fresh_dag_requests(seed):
    """A random DAG the golden files do not contain and a few random requests (label level)."""
    spec = netspec.random_dag_spec(1000 + seed, n_nodes=8 + seed % 7, cards=(2, 3, 4, 5) if seed % 2 else (4,), p_zero=0.1)
    rng = np.random.default_rng(seed)
    dom = netspec.domains(spec)
    reqs = []
    for _ in range(12):
        names = list(rng.permutation(spec["nodes"]))
        nq, ne = int(rng.integers(1, 3)), int(rng.integers(0, 4))
        q = names[:nq]
        ev = {n: dom[n][int(rng.integers(0, len(dom[n])))] for n in names[nq:nq + ne]}
        reqs.append((tuple(q), ev))
    return spec, reqs


def _oracle_series(on, q, ev):
    qs, labels, vals = on.query(list(q), ev)
    keep = vals > 0
    return qs, [l for l, k in zip(labels, keep) if k], vals[keep]


@pytest.mark.parametrize("seed", range(12))
def test_fresh_random_dags_simulator_vs_oracle(seed):
    """Beyond the golden files: new random DAGs (mixed cardinalities, exact zeros, missing rows) every seed, the
    planner's programs on the simulator against the C oracle (which the golden vectors pin to the reference)."""
    from oracle.oracle import OracleNet
    spec, reqs = _fresh_dag_requests(seed)
    on = OracleNet(spec)
    for small_cells, tiling in [(1024, (4096, 0)), (2, (4, 2))]:
        bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet), small_cells, tiling, 1)
        for (q, ev), ans in zip(reqs, bn.query_many(reqs)):
            qs, labels, vals = _oracle_series(on, q, ev)
            assert list(ans.index.names) == qs, (q, ev)
            got = [t if isinstance(t, tuple) else (t,) for t in ans.index.tolist()]
            assert got == labels, (q, ev)
            if len(vals):
                assert float(np.max(np.abs(ans.to_numpy() - vals))) <= gu.TOL, (q, ev)


def test_wide_grid_every_request_simulator_vs_oracle():
    """CPU twin of the GPU test of (almost) the same name: a 8x8 four-state grid - wide enough for CHAIN / pair / OUTER
    steps on 4^8-cell tables, small enough for the oracle - every request compared."""
    from oracle.oracle import OracleNet
    spec = netspec.grid_spec(8, 8, 4, seed=808)
    on = OracleNet(spec)
    f = flatten(netspec.build(spec, sorobn_amd.BayesNet))
    q, ev, ec = netspec.c3_requests(64, 4, 24, 3, seed=7)
    oid = np.array([on.id[f"{i:03d}"] for i in range(64)], np.int32)
    to_var = np.array([f.id[f"{i:03d}"] for i in range(64)], np.int32)
    eng = simengine.SimEngine(f)
    for i in range(24):
        codes, vals = on.query_codes([int(oid[q[i]])], oid[ev[i]].tolist(), ec[i].tolist())
        dense = np.zeros(4)
        dense[codes[:, 0]] = vals
        assert float(np.max(np.abs(eng._one([to_var[q[i]]], to_var[ev[i]], ec[i]) - dense))) <= gu.TOL


def _cache_check(f, reqs, small_cells=1024, tiling=(4096, 0)):
    """reqs: list of (qvars, evars, ecodes) id/code lists -> requests answered from a plan template."""
    L = simengine.lib()
    L.plan_sim_cache_check.restype = ctypes.c_int64
    L.plan_sim_set_small_cells(int(small_cells))
    L.plan_sim_set_tiling(int(tiling[0]), int(tiling[1]))
    q_off = np.concatenate([[0], np.cumsum([len(r[0]) for r in reqs])]).astype(np.int64)
    e_off = np.concatenate([[0], np.cumsum([len(r[1]) for r in reqs])]).astype(np.int64)
    qv = np.array([v for r in reqs for v in r[0]] or [0], np.int32)
    ev = np.array([v for r in reqs for v in r[1]] or [0], np.int32)
    ec = np.array([v for r in reqs for v in r[2]] or [0], np.int32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    rc = L.plan_sim_cache_check(ctypes.c_int32(len(f.card)), p(f.card, ctypes.c_int32), p(f.scope_off, ctypes.c_int64),
                                p(f.scope_vars, ctypes.c_int32), p(f.value_off, ctypes.c_int64), p(f.values, ctypes.c_double),
                                ctypes.c_int64(len(reqs)), p(q_off, ctypes.c_int64), p(qv, ctypes.c_int32), p(e_off, ctypes.c_int64),
                                p(ev, ctypes.c_int32), p(ec, ctypes.c_int32))
    L.plan_sim_set_small_cells(1024)
    L.plan_sim_set_tiling(4096, 0)
    assert rc >= 0, L.plan_sim_error().decode()
    return rc


def test_plan_templates_reproduce_planned_programs():
    """Requests that repeat a (query, evidence set) shape are instantiated from a template: the programs, items and
    statistics must be those of planning every request (word for word) - on Asia-style streams (config 2), on the
    10x10 grid (CHAIN / FIBER / OUTER records with evidence-sliced CPTs) and with the step forms forced onto a
    mixed-cardinality network."""
    rng = np.random.default_rng(0)

    def stream(f, n_shapes, n, max_q=1, max_e=3):
        nv = len(f.card)
        shapes = []
        for _ in range(n_shapes):
            vs = rng.permutation(nv)
            nq, ne = int(rng.integers(1, max_q + 1)), int(rng.integers(0, max_e + 1))
            shapes.append((vs[:nq].tolist(), vs[nq:nq + ne].tolist()))
        out = []
        for _ in range(n):
            q, e = shapes[int(rng.integers(0, n_shapes))]
            out.append((q, e, [int(rng.integers(0, f.card[v])) for v in e]))
        return out

    asia = flatten(netspec.build(next(n for n in _nets("examples.json") if n["spec"]["name"] == "asia")["spec"], sorobn_amd.BayesNet))
    reqs = stream(asia, 40, 1500, max_q=2)
    assert _cache_check(asia, reqs) >= 1400
    grid = flatten(netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet))
    reqs = stream(grid, 12, 120, max_e=4)
    assert _cache_check(grid, reqs) >= 100
    dag = flatten(netspec.build(_nets("wide_cards.json")[0]["spec"], sorobn_amd.BayesNet))
    reqs = stream(dag, 25, 400, max_q=2)
    assert _cache_check(dag, reqs, small_cells=2, tiling=(4, 2)) >= 350


def _device_style(f, reqs, slot_words, small_cells=1024, tiling=(4096, 0), no_prune=0):
    L = simengine.lib()
    L.plan_sim_device_style.restype = ctypes.c_int64
    L.plan_sim_set_small_cells(int(small_cells))
    L.plan_sim_set_tiling(int(tiling[0]), int(tiling[1]))
    q_off = np.concatenate([[0], np.cumsum([len(r[0]) for r in reqs])]).astype(np.int64)
    e_off = np.concatenate([[0], np.cumsum([len(r[1]) for r in reqs])]).astype(np.int64)
    qv = np.array([v for r in reqs for v in r[0]] or [0], np.int32)
    ev = np.array([v for r in reqs for v in r[1]] or [0], np.int32)
    ec = np.array([v for r in reqs for v in r[2]] or [0], np.int32)
    hints = np.ascontiguousarray(np.stack(f.hints).reshape(-1), np.int32) if len(f.hints) else np.zeros(1, np.int32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    rc = L.plan_sim_device_style(ctypes.c_int32(len(f.card)), p(f.card, ctypes.c_int32), p(f.scope_off, ctypes.c_int64),
                                 p(f.scope_vars, ctypes.c_int32), p(f.value_off, ctypes.c_int64), p(f.values, ctypes.c_double),
                                 ctypes.c_int32(len(f.hints)), p(hints, ctypes.c_int32),
                                 ctypes.c_int64(len(reqs)), p(q_off, ctypes.c_int64), p(qv, ctypes.c_int32), p(e_off, ctypes.c_int64),
                                 p(ev, ctypes.c_int32), p(ec, ctypes.c_int32), ctypes.c_int32(slot_words), ctypes.c_int32(no_prune))
    L.plan_sim_set_small_cells(1024)
    L.plan_sim_set_tiling(4096, 0)
    assert rc >= 0, L.plan_sim_error().decode()
    return rc


def test_device_planner_configuration_on_the_cpu():
    """csrc/emit_core.h the way the device runs it (emit_kernel, option gpu_emit) - the planning state of a request carved out
    of one raw, dirty slice (emit_scratch_carve), the program written into a fixed slot (EmitBuf without a ProgBuf), the
    elimination order handed over as order_search's bytes - must write plan_request's program word for word; a program
    that does not fit its slot is refused (kEmitErrWords), not truncated.  The GPU suite repeats the comparison with the
    code compiled for the device (test_device_planner_writes_the_host_programs)."""
    rng = np.random.default_rng(5)

    def stream(f, n, max_q=1, max_e=4):
        nv = len(f.card)
        out = []
        for _ in range(n):
            vs = rng.permutation(nv)
            nq, ne = int(rng.integers(1, max_q + 1)), int(rng.integers(0, max_e + 1))
            e = vs[nq:nq + ne].tolist()
            out.append((vs[:nq].tolist(), e, [int(rng.integers(0, f.card[v])) for v in e]))
        return out

    grid = flatten(netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet))
    reqs = stream(grid, 300)
    assert _device_style(grid, reqs, 6144) == len(reqs)
    fitted = _device_style(grid, reqs, 1500)            # about half of the C3 programs are longer than 1 500 - 384 words
    assert 0 < fitted < len(reqs)
    assert _device_style(grid, stream(grid, 40, max_q=2), 6144, no_prune=1) == 40
    for fname in ("random_dags.json", "wide_cards.json"):
        for net in _nets(fname):
            f = flatten(netspec.build(net["spec"], sorobn_amd.BayesNet))
            if len(f.card) > 128:
                continue
            reqs = stream(f, 60, max_q=2, max_e=3)
            assert _device_style(f, reqs, 8192) == len(reqs)
            assert _device_style(f, reqs, 8192, small_cells=3, tiling=(4, 1)) == len(reqs)


# ------------------------------------------------------------------------------------ API behaviour

@pytest.fixture()
def asia():
    spec = next(n for n in _nets("examples.json") if n["spec"]["name"] == "asia")["spec"]
    return simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))


def test_structure_attributes_match_reference_doc():
    """examples.py:255-262 (grades): nodes / children / parents."""
    bn = sorobn_amd.BayesNet(("Difficulty", "Grade"), ("Intelligence", "Grade"), ("Intelligence", "SAT"),
                             ("Grade", "Letter"))
    assert bn.nodes == ["Difficulty", "Intelligence", "Grade", "SAT", "Letter"]
    assert bn.children == {"Difficulty": ["Grade"], "Intelligence": ["Grade", "SAT"], "Grade": ["Letter"]}
    assert bn.parents == {"Grade": ["Difficulty", "Intelligence"], "SAT": ["Intelligence"], "Letter": ["Grade"]}
    bn = sorobn_amd.BayesNet(("Smoker", ["Lung cancer", "Bronchitis"]), (["Tuberculosis", "Lung cancer"], "TB or cancer"))
    assert bn.parents["TB or cancer"] == ["Lung cancer", "Tuberculosis"]
    assert bn.ancestors("TB or cancer") == {"Lung cancer", "Tuberculosis", "Smoker"}


def test_query_errors_follow_reference(asia):
    with pytest.raises(ValueError, match="At least one query variable has to be specified"):
        asia.query(event={"Smoker": True})
    with pytest.raises(ValueError, match="A query variable cannot be part of the event"):
        asia.query("Smoker", event={"Smoker": True})
    with pytest.raises(ValueError, match="Unknown algorithm, must be one of: exact, gibbs, likelihood, rejection"):
        asia.query("Smoker", event={}, algorithm="nope")
    with pytest.raises(KeyError, match="Nope"):
        asia.query("Nope", event={"Smoker": True})
    with pytest.raises(KeyError, match="Nope"):
        asia.query("Smoker", event={"Nope": True})
    with pytest.raises(TypeError):
        asia.query("Smoker")  # `event` is keyword-only and required (bayes_net.py:796-802)


def test_result_conventions(asia):
    # caller's order in the name, sorted levels in the index (bayes_net.py:869-875)
    a = asia.query("Tuberculosis", "Lung cancer", event={"Visit to Asia": True, "Smoker": True})
    assert a.name == "P(Tuberculosis, Lung cancer)" and list(a.index.names) == ["Lung cancer", "Tuberculosis"]
    assert a.to_numpy() == pytest.approx([0.855, 0.045, 0.095, 0.005], abs=1e-12)  # bayes_net.py:831-836
    # zero rows are absent
    a = asia.query("TB or cancer", "Lung cancer", event={"Tuberculosis": False})
    assert a.index.tolist() == [(False, False), (True, True)]
    assert a.to_numpy() == pytest.approx([0.945, 0.055], abs=1e-12)
    # zero-probability and out-of-domain evidence -> empty Series with the right name
    for ev in ({"TB or cancer": False, "Tuberculosis": True}, {"Smoker": "maybe"}):
        a = asia.query("Dispnea", event=ev)
        assert len(a) == 0 and a.name == "P(Dispnea)" and a.index.name == "Dispnea" and a.dtype == np.float64
    # evidence values match by equality: 1 == True
    pd.testing.assert_series_equal(asia.query("Dispnea", event={"Lung cancer": 1}),
                                   asia.query("Dispnea", event={"Lung cancer": True}))
    # single query variable -> plain Index of the domain's dtype
    a = asia.query("Dispnea", event={"Smoker": True})
    assert not isinstance(a.index, pd.MultiIndex) and a.index.dtype == bool
    # never mutates P
    before = {k: v.copy() for k, v in asia.P.items()}
    asia.query("Dispnea", event={"Smoker": True})
    for k in before:
        pd.testing.assert_series_equal(before[k], asia.P[k])


def check_reference_unit_tests_replayed(attach):
    """test_bayes_net.py:116-153, 158-226, 295-312 with the strict Series equality the reference uses; `attach(bn)` binds the backend - the CPU
    plan simulator here, nothing on the GPU box (tests/test_gpu_parity.py::test_reference_unit_tests_replayed_on_the_gpu)."""
    edges = pd.DataFrame({"parent": ["A", "B"], "child": "C"})
    bn = sorobn_amd.BayesNet(*edges.itertuples(index=False, name=None))
    bn.P["A"] = pd.Series({True: 0.7, False: 0.3})
    bn.P["B"] = pd.Series({True: 0.4, False: 0.6})
    PC = pd.DataFrame({"B": [True, True, True, True, False, False, False, False],
                       "A": [True, True, False, False, True, True, False, False],
                       "C": [True, False, True, False, True, False, True, False],
                       "p": [1, 0, 0, 1, 0.5, 0.5, 0.001, 0.999]})
    bn.P["C"] = PC.set_index(["B", "A", "C"])["p"]
    bn.prepare()
    attach(bn)
    pd.testing.assert_series_equal(bn.query("C", event={"B": False, "A": True}),
                                   pd.Series([0.5, 0.5], name="P(C)", index=pd.Index([False, True], name="C")))
    bn = sorobn_amd.BayesNet(("A", "C"), ("B", "C"))
    bn.P["A"] = pd.Series({True: 0.7, False: 0.3})
    bn.P["B"] = pd.Series({True: 0.4, False: 0.6})
    bn.P["C"] = pd.DataFrame({"A": [True, True, True, True, False, False, False, False],
                              "B": [True, True, False, False, True, True, False, False],
                              "C": [True, False, True, False, True, False, True, False],
                              "p": [1, 0, 0.5, 0.5, 0.5, 0.5, 0.001, 0.999]})
    bn.prepare()
    P = bn.P["C"]
    assert isinstance(P, pd.Series) and P.index.names == ["A", "B", "C"] and P.groupby(["A", "B"]).sum().eq(1).all()
    attach(bn)
    pd.testing.assert_series_equal(bn.query("C", event={"A": True, "B": False}),
                                   pd.Series([0.5, 0.5], name="P(C)", index=pd.Index([False, True], name="C")))
    bn = sorobn_amd.BayesNet(("Weather", "Mood"))
    bn.P["Weather"] = pd.Series({"Sunny": 0.7, "Rainy": 0.3})
    bn.P["Mood"] = pd.DataFrame({"Weather": ["Sunny", "Sunny", "Rainy", "Rainy"],
                                 "Mood": ["Happy", "Sad", "Happy", "Sad"], "p": [0.9, 0.1, 0.4, 0.6]})
    bn.prepare()
    attach(bn)
    r = bn.query("Mood", event={"Weather": "Sunny"})
    assert r["Happy"] == pytest.approx(0.9) and r["Sad"] == pytest.approx(0.1)
    # independent variables, integer labels, no structure (test_bayes_net.py:116-153)
    bn = sorobn_amd.BayesNet()
    bn.P["A"] = pd.Series({1: .2, 2: .3, 3: .5})
    bn.P["B"] = pd.Series({1: .4, 2: .2, 3: .4})
    bn.prepare()
    attach(bn)
    for k in (1, 2, 3):
        a = bn.query("A", event={"B": k})
        assert a.index.tolist() == [1, 2, 3] and a.index.dtype == np.int64
        assert a.to_numpy() == pytest.approx([.2, .3, .5], abs=1e-15)


def test_reference_unit_tests_replayed():
    check_reference_unit_tests_replayed(simengine.attach)


def test_prepare_errors_and_column_order():
    bn = sorobn_amd.BayesNet(("A", "B"))
    bn.P["A"] = pd.Series({True: 0.5, False: 0.5})
    bn.P["B"] = pd.DataFrame({"A": [True, True, False, False], "B": [True, False, True, False],
                              "prob": [0.9, 0.1, 0.4, 0.6]})
    with pytest.raises(ValueError, match="must have a 'p' column"):
        bn.prepare()
    bn.P["B"] = pd.DataFrame({"A": [True, True, False, False], "X": [True, False, True, False],
                              "p": [0.9, 0.1, 0.4, 0.6]})
    with pytest.raises(ValueError, match="has columns"):
        bn.prepare()
    mk = lambda cols: pd.DataFrame({"A": [True, True, False, False], "B": [True, False, True, False],
                                    "C": [True, True, True, True], "p": [0.9, 0.8, 0.7, 0.1]})[cols]
    nets = []
    for cols in (["A", "B", "C", "p"], ["B", "C", "A", "p"]):
        b = sorobn_amd.BayesNet(("A", "C"), ("B", "C"))
        b.P["A"] = pd.Series({True: 0.7, False: 0.3})
        b.P["B"] = pd.Series({True: 0.4, False: 0.6})
        b.P["C"] = mk(cols)
        b.prepare()
        nets.append(b)
    pd.testing.assert_series_equal(nets[0].P["C"], nets[1].P["C"])
    assert nets[0].P["C"].name == "P(C | A, B)" and nets[0].P["A"].name == "P(A)"


def test_impute_matches_reference():
    for net in _nets("impute.json"):
        bn = simengine.attach(netspec.build(net["spec"], sorobn_amd.BayesNet))
        for case in net["cases"]:
            sample = {k: v for k, v in case["sample"]}
            if "raises" in case:
                with pytest.raises(Exception):
                    bn.impute(sample)
                continue
            got = bn.impute(sample)
            pairs = [[k, netspec._py(v)] for k, v in got.items()]
            if pairs != case["expect"]:
                # only legitimate on an exact tie of the arg-max (e.g. asia: .99*.01 vs .01*.99), which
                # the reference itself resolves by last-bit rounding / PYTHONHASHSEED
                assert [k for k, _ in pairs] == [k for k, _ in case["expect"]]
                missing = [k for k, v in sample.items() if v is None]
                post = bn.query(*missing, event={k: v for k, v in sample.items() if v is not None})
                want = dict(map(tuple, case["expect"]))
                key = lambda d: tuple(d[n] for n in post.index.names)
                assert abs(post[key(want)] - post[key(dict(map(tuple, pairs)))]) < 1e-12, (net["spec"]["name"], sample)
            assert all(v is None or sample[k] == v for k, v in sample.items() if v is not None)


def test_impute_single_missing_is_sane(asia):
    got = asia.impute({"Smoker": True, "Lung cancer": None})
    assert got["Lung cancer"] == False and got["Smoker"] == True  # noqa: E712


def test_pickle_and_deepcopy_drop_device_handles(asia):
    for clone in (copy.deepcopy(asia), pickle.loads(pickle.dumps(asia))):
        assert clone._backend is None and clone.nodes == asia.nodes
        pd.testing.assert_series_equal(clone.P["Smoker"], asia.P["Smoker"])


def test_backend_rebuilds_when_cpts_change(asia):
    a = asia.query("Lung cancer", event={"Smoker": True})
    assert a.to_numpy() == pytest.approx([0.9, 0.1])
    asia.P["Lung cancer"] = pd.DataFrame({"Smoker": [True, True, False, False], "Lung cancer": [True, False, True, False],
                                          "p": [0.2, 0.8, 0.01, 0.99]})
    asia.prepare()
    simengine.attach(asia)
    assert asia.query("Lung cancer", event={"Smoker": True}).to_numpy() == pytest.approx([0.8, 0.2])


def _random_requests(ref, name, n, seed=3):
    nodes = list(ref.nodes)
    dom = netspec.domains(netspec.dump(ref, name))
    rng = np.random.default_rng(seed)
    for _ in range(n):
        perm = rng.permutation(len(nodes))
        nq = int(rng.integers(1, 3))
        ne = int(rng.integers(0, 3))
        q = [nodes[i] for i in perm[:nq]]
        ev = {nodes[i]: dom[nodes[i]][int(rng.integers(0, len(dom[nodes[i]])))] for i in perm[nq:nq + ne]}
        yield q, ev


def _reference_or_skip():
    from oracle import refload
    if not refload.available():
        pytest.skip("neither /root/reference nor oracle/_ref (make -C oracle _ref) is present")
    return refload.load()


def test_strict_series_equality_against_the_live_reference():
    """The unmodified reference is importable (sources in the build container, oracle/_ref elsewhere): compare whole
    Series objects (values, dtype, index type/dtype/names, name) with pandas' strict checker - `sorobn_amd.BayesNet`
    built from the same CPTs, answering through the CPU plan simulator."""
    ref_mod = _reference_or_skip()
    for mk in ("alarm", "asia", "sprinkler", "grades"):
        ref = getattr(ref_mod.examples, mk)()
        mine = simengine.attach(netspec.build(netspec.dump(ref, mk), sorobn_amd.BayesNet))
        for q, ev in _random_requests(ref, mk, 40):
            want = ref.query(*q, event=ev)
            got = mine.query(*q, event=ev)
            pd.testing.assert_series_equal(got, want, rtol=0, atol=1e-12, check_exact=False)
            assert type(got.index) is type(want.index) and got.index.dtype == want.index.dtype


def check_accelerated_reference_object(ref_mod, backend_factory):
    """The drop-in proper: `accelerate(ref_bn)` rebinds `_variable_elimination` / `_gibbs_sampling` / `full_joint_dist`
    of a LIVE reference object (bayes_net.py:848, 851-853, 398); its own `query` (796-875), `impute` (877-908) and
    `predict_proba` (934-962) then run unmodified on top of our backend and must return what the untouched reference
    returns.  Shared by the CPU test below (plan simulator) and the `-m gpu` test (HIP backend, reference from
    oracle/_ref)."""
    calls = {"n": 0}
    for mk in ("alarm", "asia", "sprinkler", "grades"):
        ref = getattr(ref_mod.examples, mk)()
        acc = sorobn_amd.accelerate(getattr(ref_mod.examples, mk)(), backend_factory=backend_factory)
        assert type(acc) is ref_mod.BayesNet
        inner = acc._variable_elimination

        def counted(*q, event, _inner=inner):
            calls["n"] += 1
            return _inner(*q, event=event)

        acc._variable_elimination = counted
        for q, ev in _random_requests(ref, mk, 40, seed=11):
            want = ref.query(*q, event=ev)
            before = calls["n"]
            got = acc.query(*q, event=ev)          # the reference's own query() on top of our backend
            assert calls["n"] == before + 1
            pd.testing.assert_series_equal(got, want, rtol=0, atol=1e-12, check_exact=False)
            assert type(got.index) is type(want.index) and got.index.dtype == want.index.dtype
        # impute (README.md:278-293 and friends): >= 2 missing variables - with one the reference itself fails
        nodes = list(ref.nodes)
        rng = np.random.default_rng(5)
        for _ in range(10):
            perm = rng.permutation(len(nodes))
            sample = ref.sample()
            sample = {k: (None if k in {nodes[i] for i in perm[:2]} else sample[k]) for k in nodes}
            want = ref.impute(dict(sample))
            post = ref.query(*[k for k, v in sample.items() if v is None], event={k: v for k, v in sample.items() if v is not None})
            top = np.sort(post.to_numpy())[::-1]
            if len(top) > 1 and top[0] - top[1] < 1e-9:
                continue  # an exact tie of the arg-max is resolved by last-bit rounding in the reference itself
            got = acc.impute(dict(sample))
            pd.testing.assert_series_equal(got, want)
        # full_joint_dist / predict_proba of the reference object ride on the replaced full_joint_dist
        pd.testing.assert_series_equal(acc.full_joint_dist(), ref.full_joint_dist(), rtol=0, atol=1e-12, check_exact=False)
        X = ref.sample(6)
        pd.testing.assert_series_equal(acc.predict_proba(X), ref.predict_proba(X), rtol=0, atol=1e-12, check_exact=False)
    assert calls["n"] >= 160


def check_accelerated_gibbs(ref_mod, backend_factory, n_iterations, tol):
    """`accelerate(ref_bn)._gibbs_sampling` (the seam bayes_net.py:851-853): the reference's own query(..., algorithm="gibbs")
    on a live reference object must dispatch to the rebound method, name / index the answer like the exact path does
    (869-875) and agree with the exact posterior within `tol` (one chain of n_iterations single-site updates)."""
    calls = {"n": 0}
    cases = [("sprinkler", ("Rain",), {"Sprinkler": True}), ("sprinkler", ("Wet grass", "Rain"), {"Cloudy": False}),
             ("alarm", ("Alarm", "Earthquake"), {"John calls": True}), ("alarm", ("Burglary",), {"Mary calls": True})]
    # (strictly positive networks only: on Asia's deterministic "TB or cancer" a single-site chain is not ergodic - in the
    #  reference just as here)
    for mk, q, ev in cases:
        ref = getattr(ref_mod.examples, mk)()
        acc = sorobn_amd.accelerate(getattr(ref_mod.examples, mk)(), backend_factory=backend_factory)
        inner = acc._gibbs_sampling

        def counted(*query, event, n_iterations, _inner=inner):
            calls["n"] += 1
            return _inner(*query, event=event, n_iterations=n_iterations)

        acc._gibbs_sampling = counted
        want = ref.query(*q, event=ev)
        got = acc.query(*q, event=ev, algorithm="gibbs", n_iterations=n_iterations)
        assert got.name == want.name and list(got.index.names) == list(want.index.names)
        assert type(got.index) is type(want.index) and got.dtype == np.float64
        assert abs(got.sum() - 1.0) < 1e-9
        assert set(got.index) <= set(want.index), (mk, q)  # a state of probability zero is never visited
        full = got.reindex(want.index, fill_value=0.0)
        assert float(np.max(np.abs(full.to_numpy() - want.to_numpy()))) <= tol, (mk, q, full, want)
    assert calls["n"] == len(cases)


def test_accelerate_gibbs_rebind_dispatches():
    """CPU twin of the GPU test: the rebind, the dispatch and the Series construction with a stand-in chain (the simulator's
    `gibbs` returns the exact posterior as counts - it has no chain of its own; the chain itself is a device kernel)."""
    ref_mod = _reference_or_skip()
    check_accelerated_gibbs(ref_mod, simengine.sim_backend, n_iterations=100_000, tol=1e-4)


def test_accelerate_live_reference_object_query_and_impute():
    ref_mod = _reference_or_skip()
    check_accelerated_reference_object(ref_mod, simengine.sim_backend)


def test_in_place_cpt_edit_reaches_the_backend():
    """the reference re-reads `P` on every query (bayes_net.py:770); an in-place edit of a CPT value must
    not be answered from stale device tables.  Round 5: `CptWatch` (identity of Series / index / value array + one bitwise
    comparison of all CPT numbers) instead of re-hashing every CPT per query - also for an edit through a reference the caller
    kept, a replaced CPT, renamed levels and a CPT added or removed."""
    from sorobn_amd.bayes_net import CptWatch
    spec = next(n for n in _nets("examples.json") if n["spec"]["name"] == "asia")["spec"]
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    acc = sorobn_amd.accelerate(bn, backend_factory=simengine.sim_backend)
    ev = {"Smoker": True}
    before = acc._variable_elimination("Lung cancer", event=ev).to_numpy().copy()
    w = CptWatch(bn)
    assert w.exact and not w.changed(bn)
    b0 = bn._mibn_backend()
    bn.P["Smoker"].iloc[0] += 0.0
    assert not w.changed(bn) and bn._mibn_backend() is b0          # unchanged content: no rebuild
    held = bn.P["Lung cancer"]                                      # a reference the caller keeps: the edit bypasses bn.P
    held.iloc[:] = held.to_numpy()[::-1].copy()                     # in place: same Series object, same index, same array
    assert w.changed(bn)
    after = acc._variable_elimination("Lung cancer", event=ev).to_numpy()
    assert bn._mibn_backend() is not b0 and not np.allclose(before, after)
    fresh = simengine.sim_backend(bn).variable_elimination("Lung cancer", event=ev).to_numpy()
    assert np.array_equal(after, fresh)
    for mutate in (lambda: bn.P.__setitem__("Smoker", bn.P["Smoker"].copy()),            # replaced by an equal copy
                   lambda: bn.P.__setitem__("Extra", bn.P["Smoker"].copy()),              # a CPT more
                   lambda: bn.P.pop("Extra"),                                            # a CPT less
                   lambda: setattr(bn.P["Smoker"].index, "names", ["Smoker2"])):         # levels renamed in place
        w = CptWatch(bn)
        assert not w.changed(bn)
        mutate()
        assert w.changed(bn)
    bn.P["Smoker"].index.names = ["Smoker"]
    # values that are not a numeric ndarray: the hashed fingerprint still answers
    obj = netspec.build(spec, sorobn_amd.BayesNet)
    obj.P["Smoker"] = obj.P["Smoker"].astype(object)
    w = CptWatch(obj)
    assert not w.exact and not w.changed(obj)
    obj.P["Smoker"].iloc[0] = 0.25
    assert w.changed(obj)


def test_heavy_c3_requests_simulator_vs_oracle():
    """CPU twin of tests/test_gpu_parity.py::test_c3_heavy_requests_vs_oracle (two requests): 60-120 MB plans of the C3
    stream - CHAIN / pair / OUTER steps over 4^10-cell frontiers - executed by the plan simulator against the C oracle,
    which eliminates in the planner's own order (mibn_plan_order) to finish in seconds."""
    from oracle.oracle import OracleNet
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    f = flatten(bn)
    eng = _capi.Engine(planner_only=True)
    eng.set_network(f.card, f.scope_off, f.scope_vars, f.value_off, f.values)
    if f.hints:
        eng.set_order_hints(np.stack(f.hints))
    q, ev, ec = netspec.c3_requests(100, 4, 1024, 4, seed=1)
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    cost = eng.estimate_costs(to_var[q][:, None], to_var[ev])
    heavy = [int(i) for i in np.argsort(-cost) if 5e7 < eng.plan_stats([to_var[q[i]]], to_var[ev[i]])["alg_bytes"] < 1.2e8][:2]
    assert len(heavy) == 2
    on = OracleNet(spec)
    oid = np.array([on.id[f"{i:03d}"] for i in range(100)], np.int32)
    var_to_stream = np.argsort(to_var)
    sim = simengine.SimEngine(f)
    for i in heavy:
        order = eng.plan_order([to_var[q[i]]], to_var[ev[i]])
        assert len(set(order.tolist())) == len(order) and to_var[q[i]] not in order
        prio = np.full(100, 1 << 20, np.int32)
        prio[oid[var_to_stream[order]]] = np.arange(len(order), dtype=np.int32)
        codes, vals = on.query_codes([int(oid[q[i]])], oid[ev[i]].tolist(), ec[i].tolist(), order=prio)
        got = sim._one([to_var[q[i]]], to_var[ev[i]], ec[i])
        assert float(np.max(np.abs(got[codes[:, 0]] - vals))) <= gu.TOL


def test_examples_module_and_graph_helpers_match_the_reference():
    """`sorobn_amd.examples.*` and the small graph helpers (`is_tree`, `markov_boundary`, `iter_dfs`,
    bayes_net.py:975-1075) so that code written against `sorobn.examples.asia()` ports by changing the import."""
    ref_mod = _reference_or_skip()
    for name in ("alarm", "asia", "grades", "sprinkler"):
        ref = getattr(ref_mod.examples, name)()
        mine = getattr(sorobn_amd.examples, name)()
        assert mine.nodes == ref.nodes and mine.parents == ref.parents and mine.children == ref.children
        assert mine.is_tree == ref.is_tree
        assert list(mine.iter_dfs()) == list(ref.iter_dfs())
        for node in ref.nodes:
            assert mine.markov_boundary(node) == ref.markov_boundary(node)
            a, b = mine.P[node].sort_index(), ref.P[node].sort_index()
            assert a.index.tolist() == b.index.tolist() and list(a.index.names) == list(b.index.names)
            assert np.array_equal(a.to_numpy(dtype=float), b.to_numpy(dtype=float))
    wiki = [(0, 3), (1, 4), (2, 5), (3, 6), (4, 6), (5, 8), (6, 8), (6, 9), (7, 9), (7, 10), (8, 11), (8, 12)]
    assert sorobn_amd.BayesNet(*wiki).markov_boundary(6) == [3, 4, 5, 7, 8, 9]  # bayes_net.py:1015-1031
    assert sorobn_amd.BayesNet(("a", "b"), ("a", "c")).is_tree and not sorobn_amd.BayesNet(("a", "c"), ("b", "c")).is_tree


def test_multi_request_schedule_stagger_and_threads_on_the_simulator():
    """The level-synchronous schedule of a whole chunk - items of many requests bucketed per (level, class of work),
    the workgroup -> item table, private arenas - executed by the simulator the way the level kernel sees it, with the
    invariants checked (one item of a request per level, tiles covered exactly once, bytes add up): with the requests in
    phase, with 2 / 3 / 5 staggered groups (option `stagger`) and planned by 1 or 3 workers - always the posteriors of
    the one-request-at-a-time path, bit for bit (same programs, another launch order)."""
    # (the 8x8 grid brings SWEEP items - 4^7 / 4^8-cell tables, 2-8 tiles each - whose workgroups build_schedule sizes per launch)
    for R, C, small_cells, tiling, n_req in ((6, 6, 1024, (4096, 0), 96), (6, 6, 20, (64, 2), 96), (5, 7, 3, (4, 1), 96), (8, 8, 1024, (512, 0), 40)):
        spec = netspec.grid_spec(R, C, 4, seed=R * 10 + C)
        bn = netspec.build(spec, sorobn_amd.BayesNet)
        f = flatten(bn)
        n = R * C
        sim = simengine.SimEngine(f, small_cells=small_cells, tiling=tiling)
        q, ev, ec = netspec.c3_requests(n, 4, n_req, 3, seed=2)
        to_var = np.array([f.id[f"{i:03d}"] for i in range(n)], np.int32)
        Q, E = to_var[q][:, None], to_var[ev]
        one_by_one = sim.query_fixed(Q, E, ec)
        assert np.allclose(one_by_one.sum(1), 1.0, atol=1e-12)
        for stagger, threads in ((1, 1), (2, 1), (3, 3), (5, 2)):
            got = sim.batch(Q, E, ec, stagger=stagger, threads=threads)
            assert np.array_equal(got, one_by_one), (R, C, small_cells, stagger, threads)
    # out-of-domain evidence inside a batch: that request's posterior stays all-zero, its neighbours are untouched
    ec2 = ec.copy()
    ec2[5, 0] = 7
    got = sim.batch(Q, E, ec2, stagger=2, threads=2)
    assert not got[5].any() and np.array_equal(np.delete(got, 5, 0), np.delete(one_by_one, 5, 0))


def test_sweep_form_with_other_cardinalities_around_it():
    """SWEEP steps need four-state eliminated / new / ctrl variables; the other axes of the table may have any cardinality
    (the tile takes `Rt` consecutive cells of them, ctrl values on them need a power-of-two stride).  An 8 x 9 grid whose
    outer columns have 3, 2 and 5 states: sweeps do form, and the programs - simulated the way the kernel runs them -
    agree with the CHAIN / pair programs and with the C oracle."""
    from oracle.oracle import OracleNet
    R, C = 8, 9
    spec = netspec.mixed_grid_spec(R, C, [3, 4, 4, 4, 4, 4, 4, 2, 5], seed=5)
    bn = netspec.build(spec, sorobn_amd.BayesNet)
    f = flatten(bn)
    n = R * C
    to_var = np.array([f.id[f"{i:03d}"] for i in range(n)], np.int32)
    rng = np.random.default_rng(3)
    on = OracleNet(spec)
    oid = np.array([on.id[f"{i:03d}"] for i in range(n)], np.int32)
    with_sweep, plain = simengine.SimEngine(f, tiling=(512, 0)), simengine.SimEngine(f, tiling=(512, 0))
    plain.set_option("sweep", 0)
    n_sweeps = 0
    for _ in range(16):
        qv = int(rng.integers(n - 2 * C, n))  # (a query in the last two rows: the whole grid is relevant, long sweeps)
        evs = rng.choice([v for v in range(n) if v != qv], 3, replace=False)
        codes = np.array([int(rng.integers(0, f.card[to_var[e]])) for e in evs], np.int32)
        a = with_sweep._one([to_var[qv]], to_var[evs], codes)
        bytes_sweep = with_sweep.last_stats[0]
        b = plain._one([to_var[qv]], to_var[evs], codes)
        n_sweeps += bytes_sweep < 0.97 * plain.last_stats[0]  # (a multi-variable pass replaced CHAIN / pair steps)
        assert float(np.max(np.abs(a - b))) <= 1e-14
        oc, ov = on.query_codes([int(oid[qv])], oid[evs].tolist(), codes.tolist())
        assert float(np.max(np.abs(a[oc[:, 0]] - ov))) <= gu.TOL
    assert n_sweeps >= 3, n_sweeps


def test_the_gpu_parity_streams_exercise_every_shape_of_the_sweep_tail():
    """Round 4 gave the sweep kernel a wave-owned tail for five-variable passes in which ONE digit dies (kout = 4,
    sweep_last_stage_out_dead): the digit of any of the five stages.  The GPU tests that hold that code against the register-staged
    kernel bit for bit and against the reference run the first requests of the C3 stream - this checks, on the planner's programs,
    that those requests contain canonical five-variable passes with EACH of the five digits dying, the all-survive form and the
    shapes that stay on the readout path (two dead digits)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_plan as dp
    L = simengine.lib()  # (the simulator's planner options are process-wide: the product's defaults, whatever an earlier test left)
    L.plan_sim_set_small_cells(1024); L.plan_sim_set_tiling(4096, 0); L.plan_sim_set_fuse(1); L.plan_sim_set_chain(1)
    L.plan_sim_set_sweep(5); L.plan_sim_set_sweep_min(2); L.plan_sim_set_prune(1)
    f = flatten(netspec.build(netspec.grid_spec(10, 10, 4, seed=0), sorobn_amd.BayesNet))
    to_var = np.array([f.id[f"{i:03d}"] for i in range(100)], np.int32)
    q, ev, ec = netspec.c3_requests(100, 4, 300, 4, seed=1)
    own, dead, two_dead = 0, set(), 0
    packs = {0x4321: 0, 0x4320: 1, 0x4310: 2, 0x4210: 3, 0x3210: 4}
    for i in range(300):
        w = dp.program(f, [to_var[q[i]]], to_var[ev[i]], ec[i])
        p = 1
        for s in dp.decode(w):
            if s["kind"] == "SWEEP" and s["k"] == 5 and (int(w[p + 1]) >> 16) & 16:  # canonical five-variable pass
                surv = int(w[p + 8])
                died = [4 - j for j, g in enumerate(s["stages"]) if g["cout"] == 1]
                if s["kout"] == 5:
                    assert surv == 0x43210 and not died
                    own += 1
                elif s["kout"] == 4:
                    assert len(died) == 1 and packs[surv] == died[0], (hex(surv), died)  # the survivors stay in place, ascending
                    dead.add(died[0])
                elif s["kout"] == 3:
                    two_dead += 1
            p += s["words"]
    assert own > 100 and dead == {0, 1, 2, 3, 4} and two_dead > 0, (own, dead, two_dead)


# ------------------------------------------------------------------------------------ round 5: the pandas boundary, vectorised

def check_pandas_batch_api(attach):
    """Shared by the CPU test below (plan simulator) and tests/test_gpu_parity.py (HIP backend): `query()` (finished Series built
    directly, Backend._tail), `query_many` (-> PosteriorBatch: lazily built Series, bulk-encoded, pipelined) and its `to_frame()`,
    and `query_frame` - every answer strictly equal (pandas' checker: values, dtype, index class / dtype / names / level order /
    row order, name) to the tail of the reference's `query` (bayes_net.py:869-875) applied to the unfinished posterior."""
    n_checked = 0
    for fname, take in (("examples.json", 120), ("random_dags.json", 30), ("wide_cards.json", 16)):
        for net in _nets(fname):
            bn = attach(netspec.build(net["spec"], sorobn_amd.BayesNet))
            reqs = [(tuple(r["query"]), {k: v for k, v in r["event"]}) for r in net["requests"][:take]]
            batch = bn.query_many(reqs)
            assert len(batch) == len(reqs) and len(batch[1:3]) == 2
            for i, (q, e) in enumerate(reqs):
                want = bn._finish(bn.backend.variable_elimination(*q, event=e), q)  # rename / reorder_levels / sort_index
                pd.testing.assert_series_equal(batch[i], want, check_exact=True)
                pd.testing.assert_series_equal(bn.query(*q, event=e), want, check_exact=True)
                assert type(batch[i].index) is type(want.index)
                n_checked += 1
            frame = batch.to_frame()  # mixed query variables: one row per positive cell
            assert isinstance(frame, pd.DataFrame) and frame["p"].gt(0).all()
            assert frame.groupby("request")["p"].sum().round(9).isin([1.0]).all()  # every non-empty answer is a distribution
            assert int(frame["request"].nunique()) == sum(len(batch[i]) > 0 for i in range(len(batch)))
    # one query tuple for the whole batch: to_frame() is a long Series (request, *sorted names) whose slices ARE the answers
    spec = next(n for n in _nets("examples.json") if n["spec"]["name"] == "asia")["spec"]
    bn = attach(netspec.build(spec, sorobn_amd.BayesNet))
    for q in (("Lung cancer",), ("Tuberculosis", "Lung cancer"), ("Smoker", "Bronchitis", "Dispnea")):
        evs = [{}, {"Visit to Asia": True}, {"Visit to Asia": False, "Positive X-ray": True}, {"Positive X-ray": False}]
        evs = [{k: v for k, v in e.items() if k not in q} for e in evs]
        batch = bn.query_many([(q, e) for e in evs])
        long = batch.to_frame()
        assert isinstance(long, pd.Series) and list(long.index.names) == ["request", *sorted(q)]
        for i in range(len(evs)):
            got = long.xs(i, level="request")
            if len(q) > 1:
                got.index = got.index.remove_unused_levels()
            want = batch[i]
            want2 = want.copy()
            if len(q) > 1:
                want2.index = want2.index.remove_unused_levels()
            pd.testing.assert_series_equal(got, want2, check_names=False, check_exact=True)
            assert list(got.index.names) == list(want.index.names)
        # query_frame: events as rows, NaN = unobserved; row r = the dense posterior in the order of query()'s index
        cols = sorted({k for e in evs for k in e})
        X = pd.DataFrame([{c: e.get(c, np.nan) for c in cols} for e in evs], index=[f"r{i}" for i in range(len(evs))]).astype(object)
        out = bn.query_frame(*q, events=X)
        assert list(out.index) == list(X.index) and out.shape[0] == len(evs)
        for i in range(len(evs)):
            row = out.iloc[i]
            want = batch[i]
            got = row[row > 0]
            assert np.array_equal(got.to_numpy(), want.to_numpy())
            assert [k if isinstance(k, tuple) else (k,) for k in got.index.tolist()] == [k if isinstance(k, tuple) else (k,) for k in want.index.tolist()]
            assert list(out.columns.names) == list(want.index.names)
    with pytest.raises(ValueError):
        bn.query_frame("Smoker", events=pd.DataFrame({"Smoker": [True]}))
    with pytest.raises(ValueError):
        bn.query_many([(("Smoker",), {"Smoker": True})])
    with pytest.raises(KeyError):
        bn.query_many([(("Smoker",), {}), (("No such node",), {})])
    assert len(bn.query_many([])) == 0
    return n_checked


def test_pandas_batch_api_on_the_simulator():
    assert check_pandas_batch_api(simengine.attach) > 500


def test_bulk_encode_matches_encode():
    """Backend.encode_many (dict lookups in bulk) against Backend.encode request by request: same CSR arrays, incl. int-for-bool and
    out-of-domain labels (code -1 through the per-request path), and the reference's KeyError for unknown names."""
    for net in _nets("examples.json"):
        bn = simengine.attach(netspec.build(net["spec"], sorobn_amd.BayesNet))
        be = bn.backend
        reqs = [(tuple(r["query"]), {k: v for k, v in r["event"]}) for r in net["requests"]]
        q_off, qv, e_off, ev, ec = be.encode_many(reqs)
        qs, es, cs = [], [], []
        for q, e in reqs:
            a, b, c = be.encode(q, e)
            qs += a
            es += b
            cs += c
        assert qv.tolist() == qs and ev.tolist() == es and ec.tolist() == cs
        assert q_off.tolist() == np.concatenate([[0], np.cumsum([len(q) for q, _ in reqs])]).tolist()
        assert e_off.tolist() == np.concatenate([[0], np.cumsum([len(e) for _, e in reqs])]).tolist()


def test_planner_output_is_pinned(tmp_path):
    """The planner's optimisations of rounds 4 and 5 (static slot sets, integer min-fill keys, the small-step shortcut, popcount scope
    sizes) must not move a word: tools/plan_fingerprint.cpp plans seeded request streams on 12 synthetic networks x 6 option sets and
    prints one hash of programs + work items + statistics + schedule per pair; tools/plan_fingerprint.expected.txt is that output for
    the committed planner.  (A deliberate change of the planner's choices or of the class order re-pins the file.)"""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "plan_fingerprint"
    r = subprocess.run(["g++", "-O2", "-mpopcnt", "-std=c++17", os.path.join(ROOT, "tools", "plan_fingerprint.cpp"),
                        os.path.join(ROOT, "sorobn_amd", "csrc", "planner.cpp"), "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600).stdout
    want = open(os.path.join(ROOT, "tools", "plan_fingerprint.expected.txt")).read()
    assert out.splitlines() == want.splitlines()


def test_query_many_pipeline_on_the_simulator():
    """Backend.exact_many's long-batch path (sub-batches checked / encoded on a helper thread, two engine calls in flight) with the plan
    simulator behind a submit_fixed / wait pair: the same answers as the one-call path, a ragged batch falls back to one CSR call, a
    malformed or unknown request raises on the caller's thread after the call in flight has been collected."""
    spec = netspec.grid_spec(4, 4, 3, seed=2)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    eng = bn.backend.engine
    waited = []
    eng.submit_fixed = lambda q, e, c: eng.query_fixed(q, e, c)
    eng.wait = lambda h: (waited.append(len(h)), h)[1]
    q, ev, ec = netspec.c3_requests(16, 3, 70, 2, seed=4)
    reqs = [((f"{a:03d}",), {f"{v:03d}": int(c) for v, c in zip(vs, cs)}) for a, vs, cs in zip(q.tolist(), ev.tolist(), ec.tolist())]
    one_call = bn.query_many(reqs)                      # 70 requests <= the default sub-batch: one call
    assert not waited
    piped = bn.query_many(reqs, sub_batch=16)           # five sub-batches through the pipeline
    assert waited == [16, 16, 16, 16, 6] and np.array_equal(piped.out, one_call.out) and np.array_equal(piped.out_off, one_call.out_off)
    for i in (0, 15, 16, 69):
        pd.testing.assert_series_equal(piped[i], bn.query(*reqs[i][0], event=reqs[i][1]), check_exact=True)
    bare = [(r[0][0], r[1]) for r in reqs]              # bare names instead of 1-tuples
    assert np.array_equal(bn.query_many(bare, sub_batch=16).out, one_call.out)
    ragged = list(reqs)
    ragged[40] = (("000", "015"), {"005": 1})           # another arity in the third sub-batch: one CSR call for everything
    del waited[:]
    rg = bn.query_many(ragged, sub_batch=16)
    assert len(rg) == 70 and len(rg.dense(40)) == 9 and np.array_equal(rg.dense(41), one_call.dense(41))
    pd.testing.assert_series_equal(rg[40], bn.query("000", "015", event={"005": 1}), check_exact=True)
    for bad, exc in (((("no such node",), {"000": 0}), KeyError), ((("003",), {"003": 1}), ValueError), (((), {"003": 1}), ValueError)):
        broken = list(reqs)
        broken[50] = bad
        with pytest.raises(exc):
            bn.query_many(broken, sub_batch=16)
    assert np.array_equal(bn.query_many(reqs, sub_batch=16).out, one_call.out)  # the backend is as usable as before


def test_query_many_pipeline_mixed_query_cardinalities():
    """a long fixed-arity batch whose query variable alternates between a 2-state and a 3-state node.  The pipelined
    path used to reshape every sub-batch's posteriors to [B, cells]; now an engine with `wait_flat` (the real one) returns them back to
    back, and an engine without it (a test double) takes the batch through the general CSR call.  Both against the one-call path."""
    rng = np.random.default_rng(5)
    bn = sorobn_amd.BayesNet(("a", "b"), ("a", "c"), ("b", "d"), ("c", "d"))
    bn.P["a"] = pd.Series([0.3, 0.7], index=[0, 1])
    bn.P["b"] = pd.Series(rng.dirichlet(np.ones(3), size=2).reshape(-1), index=pd.MultiIndex.from_product([[0, 1], [0, 1, 2]], names=["a", "b"]))
    bn.P["c"] = pd.Series(rng.dirichlet(np.ones(2), size=2).reshape(-1), index=pd.MultiIndex.from_product([[0, 1], [0, 1]], names=["a", "c"]))
    bn.P["d"] = pd.Series(rng.dirichlet(np.ones(3), size=6).reshape(-1),
                          index=pd.MultiIndex.from_product([[0, 1, 2], [0, 1], [0, 1, 2]], names=["b", "c", "d"]))
    bn.prepare()
    bn = simengine.attach(bn)
    eng = bn.backend.engine
    reqs = [(("c",) if i % 2 else ("b",), {"a": int(i % 2)}) for i in range(22)]  # cells 3, 2, 3, 2, ...: 5 per pair - not divisible by a sub-batch of 4
    one_call = bn.query_many(reqs)
    assert [len(one_call.dense(i)) for i in range(4)] == [3, 2, 3, 2]

    def submit(q, e, c):  # what the real engine does: the posteriors of a sub-batch back to back
        B = len(q)
        out, off = eng.query_batch(np.arange(B + 1) * q.shape[1], q.reshape(-1), np.arange(B + 1) * e.shape[1], e.reshape(-1), c.reshape(-1))
        return out
    eng.submit_fixed = submit
    flat_calls = []
    eng.wait_flat = lambda h: (flat_calls.append(len(h)), h)[1]
    eng.wait = lambda h: h
    piped = bn.query_many(reqs, sub_batch=4)
    assert flat_calls == [10, 10, 10, 10, 10, 5] and np.array_equal(piped.out, one_call.out) and np.array_equal(piped.out_off, one_call.out_off)
    for i in (0, 1, 21):
        pd.testing.assert_series_equal(piped[i], bn.query(*reqs[i][0], event=reqs[i][1]), check_exact=True)
    del eng.wait_flat  # an engine without the flat wait: the mixed batch goes as one CSR call
    eng.submit_fixed = lambda q, e, c: eng.query_fixed(q, e, c)
    again = bn.query_many(reqs, sub_batch=4)
    assert np.array_equal(again.out, one_call.out) and np.array_equal(again.out_off, one_call.out_off)


def test_answers_do_not_alias_the_caches():
    """renaming the index of an answer must not reach the cached tail of the next answer, and writing into a Series
    taken from a PosteriorBatch must not reach the batch's buffer."""
    spec = netspec.grid_spec(3, 3, 3, seed=1)
    bn = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    ans = bn.query("004", event={"000": 1})
    ans.index.name = "renamed"
    assert bn.query("004", event={"000": 1}).index.name == "004"
    batch = bn.query_many([(("004",), {"000": 1}), (("005",), {"001": 2})])
    s = batch[0]
    before = batch.dense(0).copy()
    s.iloc[0] = 99.0
    assert np.array_equal(batch.dense(0), before) and batch[0].iloc[0] == before[0]


def test_subclass_override_of_the_dispatch_hook_is_honoured():
    """`_variable_elimination` is the hook the reference's query() dispatches to (bayes_net.py:848); the exact fast
    path of query() must step aside for a subclass that overrides it at class level, not only for an instance-level rebind."""
    calls = []

    class Traced(sorobn_amd.BayesNet):
        def _variable_elimination(self, *query, event):
            calls.append(query)
            return super()._variable_elimination(*query, event=event)

    spec = netspec.grid_spec(3, 3, 3, seed=1)
    bn = simengine.attach(netspec.build(spec, Traced))
    plain = simengine.attach(netspec.build(spec, sorobn_amd.BayesNet))
    pd.testing.assert_series_equal(bn.query("004", event={"000": 1}), plain.query("004", event={"000": 1}), check_exact=True)
    assert calls == [("004",)]
