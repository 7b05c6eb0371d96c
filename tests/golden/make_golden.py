"""Generate golden vectors from the UNMODIFIED reference (MaxHalford/sorobn @ /root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONHASHSEED=0 python tests/golden/make_golden.py [--heavy]

Writes tests/golden/*.json.  Every expected value is `BayesNet.query(...)` /
`BayesNet.impute(...)` of the reference (bayes_net.py:796-908) evaluated here; values are stored as
float.hex() so they round-trip bit-exactly, the index is stored as label tuples and compared
exactly by the tests.  The reference needs `vose` only for sampling; oracle/refload.py registers a
stand-in (exact path never calls it).  Synthetic networks use hash-ordered node names so the
reference's set-iteration elimination order (bayes_net.py:766,779) is ascending node id.
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import refload  # noqa: E402
import netspec  # noqa: E402

assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
sorobn = refload.load()


def series_to_json(s):
    idx = s.index
    names = [str(n) for n in idx.names]
    rows = [list(map(netspec._py, k if isinstance(k, tuple) else (k,))) for k in idx.tolist()]
    return {"name": s.name, "index_names": names, "index": rows,
            "multi": isinstance(idx, pd.MultiIndex),
            "values_hex": [float(v).hex() for v in s.tolist()]}


def run_requests(bn, reqs, wrap):
    out = []
    for q, ev in reqs:
        t0 = time.time()
        ans = bn.query(*[wrap(x) for x in q], event={wrap(k): v for k, v in ev})
        out.append({"query": list(q), "event": [[k, netspec._py(v)] for k, v in ev],
                    "expect": series_to_json(ans), "ref_seconds": round(time.time() - t0, 4)})
    return out


def all_requests(spec, max_q=1, max_e=3, limit=None, seed=0):
    dom = netspec.domains(spec)
    nodes = spec["nodes"]
    reqs = []
    for nq in range(1, max_q + 1):
        for q in itertools.combinations(nodes, nq):
            rest = [n for n in nodes if n not in q]
            for ne in range(0, max_e + 1):
                for es in itertools.combinations(rest, ne):
                    for vals in itertools.product(*[dom[e] for e in es]):
                        reqs.append((q, list(zip(es, vals))))
    if limit and len(reqs) > limit:
        rng = np.random.default_rng(seed)
        keep = sorted(rng.choice(len(reqs), size=limit, replace=False).tolist())
        reqs = [reqs[i] for i in keep]
    return reqs


def random_requests(spec, n, seed, max_q=2, max_e=4):
    rng = np.random.default_rng(seed)
    dom = netspec.domains(spec)
    nodes = spec["nodes"]
    reqs = []
    for _ in range(n):
        nq = int(rng.integers(1, max_q + 1))
        ne = int(rng.integers(0, min(max_e, len(nodes) - nq) + 1))
        perm = rng.permutation(len(nodes))
        q = tuple(nodes[i] for i in perm[:nq])
        es = [nodes[i] for i in perm[nq:nq + ne]]
        ev = [(e, dom[e][int(rng.integers(0, len(dom[e])))]) for e in es]
        reqs.append((q, ev))
    return reqs


def examples_fixture():
    nets = []
    mk = {"alarm": sorobn.examples.alarm, "asia": sorobn.examples.asia,
          "sprinkler": sorobn.examples.sprinkler, "grades": sorobn.examples.grades}
    for name, fn in mk.items():
        bn = fn()
        spec = netspec.dump(bn, name)
        reqs = all_requests(spec, max_q=1, max_e=3, limit=400, seed=1)
        reqs += all_requests(spec, max_q=2, max_e=2, limit=150, seed=2)[:150]
        # edge cases: out-of-domain evidence value, int-for-bool evidence (1 == True)
        reqs.append(((spec["nodes"][0],), [(spec["nodes"][-1], "no-such-label")]))
        if name in ("alarm", "asia", "sprinkler"):
            reqs.append(((spec["nodes"][0],), [(spec["nodes"][-1], 1)]))
        nets.append({"spec": spec, "requests": run_requests(bn, reqs, lambda s: s)})
        print(name, len(reqs), flush=True)

    # networks of the reference's unit tests (test_bayes_net.py:86-91,116-153,158-226,295-312)
    bn = sorobn.BayesNet("A", "B", "C")
    bn.P["A"] = pd.Series({True: 0.1, False: 0.9})
    bn.P["B"] = pd.Series({True: 0.3, False: 0.7})
    bn.P["C"] = pd.Series({True: 0.5, False: 0.5})
    bn.prepare()
    spec = netspec.dump(bn, "naive")
    nets.append({"spec": spec, "requests": run_requests(bn, all_requests(spec, 2, 2), lambda s: s)})

    bn = sorobn.BayesNet()
    bn.P["A"] = pd.Series({1: 0.2, 2: 0.3, 3: 0.5})
    bn.P["B"] = pd.Series({1: 0.4, 2: 0.2, 3: 0.4})
    bn.prepare()
    spec = netspec.dump(bn, "indep_int_labels")
    spec["nodes"] = ["A", "B"]
    nets.append({"spec": spec, "requests": run_requests(bn, all_requests(spec, 2, 1), lambda s: s)})

    bn = sorobn.BayesNet(("A", "C"), ("B", "C"))
    bn.P["A"] = pd.Series({True: 0.7, False: 0.3})
    bn.P["B"] = pd.Series({True: 0.4, False: 0.6})
    PC = pd.DataFrame({"B": [True, True, True, True, False, False, False, False],
                       "A": [True, True, False, False, True, True, False, False],
                       "C": [True, False, True, False, True, False, True, False],
                       "p": [1, 0, 0, 1, 0.5, 0.5, 0.001, 0.999]})
    bn.P["C"] = PC.set_index(["B", "A", "C"])["p"]
    bn.prepare()
    spec = netspec.dump(bn, "issue19_index_names")
    nets.append({"spec": spec, "requests": run_requests(bn, all_requests(spec, 2, 2), lambda s: s)})

    bn = sorobn.BayesNet(("Weather", "Mood"))
    bn.P["Weather"] = pd.Series({"Sunny": 0.7, "Rainy": 0.3})
    bn.P["Mood"] = pd.DataFrame({"Weather": ["Sunny", "Sunny", "Rainy", "Rainy"],
                                 "Mood": ["Happy", "Sad", "Happy", "Sad"],
                                 "p": [0.9, 0.1, 0.4, 0.6]})
    bn.prepare()
    spec = netspec.dump(bn, "string_labels")
    nets.append({"spec": spec, "requests": run_requests(bn, all_requests(spec, 2, 1), lambda s: s)})
    netspec.save(os.path.join(HERE, "examples.json"), nets)


def impute_fixture():
    out = []
    for name, fn in {"alarm": sorobn.examples.alarm, "asia": sorobn.examples.asia,
                     "grades": sorobn.examples.grades}.items():
        bn = fn()
        spec = netspec.dump(bn, name)
        dom = netspec.domains(spec)
        rng = np.random.default_rng(7)
        cases = []
        for _ in range(25):
            nodes = spec["nodes"]
            nmiss = int(rng.integers(2, min(4, len(nodes)) + 1))
            perm = rng.permutation(len(nodes))
            sample = {}
            for j, i in enumerate(perm):
                n = nodes[i]
                sample[n] = None if j < nmiss else dom[n][int(rng.integers(0, len(dom[n])))]
            # keep dict order random but deterministic
            try:
                res = bn.impute(sample)
            except Exception as e:  # zero-probability evidence -> empty posterior -> idxmax raises
                cases.append({"sample": [[k, netspec._py(v) if v is not None else None]
                                         for k, v in sample.items()],
                              "raises": type(e).__name__})
                continue
            cases.append({"sample": [[k, netspec._py(v) if v is not None else None]
                                     for k, v in sample.items()],
                          "expect": [[str(k), netspec._py(v)] for k, v in res.items()]})
        out.append({"spec": spec, "cases": cases})
    netspec.save(os.path.join(HERE, "impute.json"), out)


def joint_fixture():
    """full_joint_dist (both keep_zeros settings), predict_proba / predict_log_proba for
    DataFrames over all / some / one column and for dicts (bayes_net.py:398-465, 934-973)."""
    out = []
    nets = {"alarm": sorobn.examples.alarm(), "asia": sorobn.examples.asia(), "sprinkler": sorobn.examples.sprinkler(),
            "grades": sorobn.examples.grades()}
    for seed in (0, 3, 5):  # sparse CPTs with zeros and missing rows
        spec = netspec.random_dag_spec(seed, n_nodes=6, labels="str" if seed == 3 else "int")
        nets[f"dag{seed}"] = (spec, netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName))
    for name, bn in nets.items():
        if isinstance(bn, tuple):
            spec, bn = bn
            spec = dict(spec, name=name)
        else:
            spec = netspec.dump(bn, name)
        entry = {"spec": spec, "fjd": series_to_json(bn.full_joint_dist()),
                 "fjd_keep_zeros": series_to_json(bn.full_joint_dist(keep_zeros=True)), "predict": []}
        fjd = bn.full_joint_dist()
        names = [str(n) for n in fjd.index.names]
        rng = np.random.default_rng(11)
        rows = fjd.index.to_frame(index=False).iloc[rng.integers(0, len(fjd), 12)].reset_index(drop=True)
        rows.columns = names
        wrapn = (lambda n: refload.HashedName(n)) if name.startswith("dag") else (lambda n: n)
        for cols in (names, names[:2], names[1:4], names[-1:]):
            X = rows[list(cols)].copy()
            X.columns = [wrapn(c) for c in cols]
            res = bn.predict_proba(X)
            lres = bn.predict_log_proba(X)
            entry["predict"].append({"columns": list(cols), "rows": [[netspec._py(v) for v in r] for r in X.values.tolist()],
                                     "expect": series_to_json(res),
                                     "log_values_hex": [float(v).hex() for v in lres.tolist()]})
        d = {wrapn(k): v for k, v in zip(names, rows.iloc[0].tolist())}
        entry["predict_dict"] = {"sample": [[str(k), netspec._py(v)] for k, v in d.items()],
                                 "expect_hex": float(bn.predict_proba(d)).hex()}
        out.append(entry)
        print("joint", name, len(fjd), flush=True)
    netspec.save(os.path.join(HERE, "joint.json"), out)


def learn_fixture():
    """fit / partial_fit (bayes_net.py:467-516) and structure.chow_liu
    (structure.py:9-63) on fixed data sets (forward samples of the example networks drawn with numpy)."""
    out = []
    for name, fn in {"alarm": sorobn.examples.alarm, "asia": sorobn.examples.asia, "grades": sorobn.examples.grades}.items():
        bn = fn()
        spec = netspec.dump(bn, name)
        fjd = bn.full_joint_dist()
        rng = np.random.default_rng(21)
        rows = fjd.index.to_frame(index=False).iloc[rng.choice(len(fjd), size=400, p=fjd.to_numpy())].reset_index(drop=True)
        rows.columns = [str(c) for c in fjd.index.names]
        entry = {"spec": spec, "columns": list(rows.columns), "rows": [[netspec._py(v) for v in r] for r in rows.values.tolist()]}
        for prior in (None, 1):
            b = fn()
            b.prior_count = prior
            b.fit(rows)
            entry[f"fit_prior_{prior}"] = {n: series_to_json(b.P[n].sort_index()) for n in b.P}
        b = fn()
        b.P = {}
        b._P_sizes = {}
        for chunk in np.array_split(rows, 5):
            b.partial_fit(chunk)
        entry["partial_fit_5"] = {n: series_to_json(b.P[n].sort_index()) for n in b.P}
        edges = sorobn.structure.chow_liu(rows)
        entry["chow_liu"] = [[str(u), str(v)] for u, v in edges]
        entry["chow_liu_root_last"] = [[str(u), str(v)] for u, v in sorobn.structure.chow_liu(rows, root=rows.columns[-1])]
        marg = {v: rows[v].value_counts(normalize=True) for v in rows.columns}
        entry["mutual_info"] = [[u, v, float(sorobn.structure.mutual_info(rows.groupby([u, v]).size() / len(rows), marg[u], marg[v])).hex()]
                                for u, v in itertools.combinations(sorted(rows.columns), 2)]
        out.append(entry)
        print("learn", name, len(rows), flush=True)
    netspec.save(os.path.join(HERE, "learn.json"), out)


def dags_fixture():
    nets = []
    for seed in range(24):
        spec = netspec.random_dag_spec(seed, labels="str" if seed % 3 == 2 else "int")
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        reqs = random_requests(spec, 24, seed=1000 + seed)
        nets.append({"spec": spec, "requests": run_requests(bn, reqs, refload.HashedName)})
        print("dag", seed, len(spec["nodes"]), flush=True)
    netspec.save(os.path.join(HERE, "random_dags.json"), nets)


def wide_fixture():
    """Higher cardinalities than the other fixtures (the kernels' runtime-cx / runtime-NC classes, joint
    eliminations of unequal cards, cx up to 16 from one variable): random DAGs with cards up to 11 and small
    K = 8 / K = 16 grids."""
    nets = []
    for seed in range(8):
        spec = netspec.random_dag_spec(100 + seed, n_nodes=6 + seed % 4, cards=(2, 3, 6, 8, 11),
                                       labels="str" if seed % 2 else "int")
        spec["name"] = f"wide_dag{seed}"
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        reqs = random_requests(spec, 16, seed=2000 + seed, max_q=2, max_e=3)
        nets.append({"spec": spec, "requests": run_requests(bn, reqs, refload.HashedName)})
        print("wide dag", seed, len(spec["nodes"]), flush=True)
    for R, C, K in [(3, 3, 8), (3, 4, 8), (2, 3, 16), (2, 5, 7)]:
        spec = netspec.grid_spec(R, C, K, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        n = R * C
        reqs = [((f"{0:03d}",), [(f"{n - 1:03d}", 0)]), ((f"{n - 1:03d}",), [(f"{0:03d}", K - 1)])]
        reqs += random_requests(spec, 8, seed=R * 100 + C * 10 + K, max_q=2, max_e=3)
        nets.append({"spec": spec, "requests": run_requests(bn, reqs, refload.HashedName)})
        print("wide grid", R, C, K, flush=True)
    netspec.save(os.path.join(HERE, "wide_cards.json"), nets)


def huge_fixture():
    """Cardinalities far above the kernels' compile-time shapes (the reference's join has no limit,
    bayes_net.py:233-250): random DAGs of 5-8 nodes whose cardinalities come from {2, 3, 17, 33, 64, 100} - axes of 17 ... 100
    states in the GENERIC / FIBER runtime-cx / runtime-NC kernels - with CPTs capped at 12 000 cells (the recipe is stored, not the
    CPTs: golden_util.dag_spec_from_recipe), zeros and missing rows as in the other DAG fixtures; plus chains / a 2x2 grid of one
    huge cardinality."""
    out = []
    for k, (seed, n, cards) in enumerate([(300, 5, (17, 33, 64, 100)), (301, 6, (2, 3, 17, 33, 64, 100)), (302, 7, (3, 17, 100)),
                                          (303, 8, (2, 17, 33, 64)), (304, 6, (33, 100)), (305, 5, (64, 100))]):
        recipe = {"name": f"huge_dag{k}", "seed": seed, "n_nodes": n, "cards": list(cards), "max_parents": 2, "max_cells": 12000,
                  "labels": "str" if k % 2 else "int"}
        entry = {"recipe": recipe}
        r = dict(recipe)
        name = r.pop("name")
        spec = netspec.random_dag_spec(r.pop("seed"), cards=tuple(r.pop("cards")), **r)
        spec["name"] = name
        entry["cpt_sum_hex"] = float(sum(row[-1] for c in spec["cpts"].values() for row in c["rows"])).hex()
        entry["cards"] = {nm: len(d) for nm, d in netspec.domains(spec).items()}
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        reqs = random_requests(spec, 12, seed=3000 + k, max_q=1, max_e=3) + random_requests(spec, 2, seed=3100 + k, max_q=2, max_e=2)
        entry["requests"] = run_requests(bn, reqs, refload.HashedName)
        out.append(entry)
        print("huge dag", k, entry["cards"], [r["ref_seconds"] for r in entry["requests"]], flush=True)
    netspec.save(os.path.join(HERE, "huge_cards.json"), out)


def many_nodes_fixture():
    """Networks with more than 128 variables: the planner's generic (kMaxVars-wide) bitset paths instead of the
    two-word fast paths."""
    nets = []
    for R, C, K in [(12, 12, 2), (50, 3, 3), (90, 2, 4)]:
        spec = netspec.grid_spec(R, C, K, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        n = R * C
        reqs = [((f"{0:03d}",), [(f"{n - 1:03d}", 0)]), ((f"{n - 1:03d}",), [(f"{0:03d}", K - 1)])]
        reqs += random_requests(spec, 8, seed=R * 100 + C * 10 + K, max_q=2, max_e=4)
        nets.append({"spec": spec, "requests": run_requests(bn, reqs, refload.HashedName)})
        print("many-nodes grid", R, C, K, [r["ref_seconds"] for r in nets[-1]["requests"]], flush=True)
    netspec.save(os.path.join(HERE, "many_nodes.json"), nets)


def grids_fixture(heavy):
    nets = []
    small = [(2, 2, 2), (2, 3, 3), (3, 3, 4), (3, 4, 2), (4, 4, 4), (4, 5, 3), (5, 5, 4), (6, 6, 4),
             (5, 10, 2), (6, 6, 3)]
    for R, C, K in small:
        spec = netspec.grid_spec(R, C, K, seed=0)
        bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
        n = R * C
        reqs = [((f"{0:03d}",), [(f"{n - 1:03d}", 0)]), ((f"{n - 1:03d}",), [(f"{0:03d}", 0)])]
        reqs += random_requests(spec, 10, seed=R * 100 + C * 10 + K, max_q=2, max_e=4)
        nets.append({"recipe": {"R": R, "C": C, "K": K, "seed": 0},
                     "cpt_sum_hex": float(sum(r[-1] for c in spec["cpts"].values()
                                              for r in c["rows"])).hex(),
                     "requests": run_requests(bn, reqs, refload.HashedName)})
        print("grid", R, C, K, flush=True)
    netspec.save(os.path.join(HERE, "grids_small.json"), nets)

    # 10x10 K=4: the BASELINE C3 network.  First requests of the C3 stream whose reference cost is
    # modest, plus the SURVEY Appendix A "typical" request; --heavy adds the 448 s worst case.
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
    reqs = [(("055",), [("000", 0), ("009", 0), ("090", 0)])]
    q, ev, ec = netspec.c3_requests(100, 4, 400, 4, seed=1)
    picked = 0
    for i in range(400):
        special = [int(q[i])] + ev[i].tolist()
        # scope-only cost proxy: keep requests whose staircase is narrow (reference finishes in s)
        rows = [s // 10 for s in special]
        cols = [s % 10 for s in special]
        ext = [max([c for r, c in zip(rows, cols) if r >= rr], default=-1) + 1 for rr in range(10)]
        if max(ext) <= 8 and picked < 40:
            reqs.append(((f"{q[i]:03d}",), [(f"{e:03d}", int(c)) for e, c in zip(ev[i], ec[i])]))
            picked += 1
    # a few full-width ones (tens of seconds each in the reference)
    reqs.append((("099",), [("090", 1), ("009", 2), ("045", 3), ("054", 0)]))
    reqs.append((("037", "062"), [("091", 1), ("019", 2)]))
    if heavy:
        reqs.append((("099",), [("000", 0)]))
    t0 = time.time()
    res = run_requests(bn, reqs, refload.HashedName)
    print("grid10x10", len(reqs), round(time.time() - t0, 1), "s", flush=True)
    netspec.save(os.path.join(HERE, "grid10x10.json"),
                 {"recipe": {"R": 10, "C": 10, "K": 4, "seed": 0},
                  "cpt_sum_hex": float(sum(r[-1] for c in spec["cpts"].values()
                                           for r in c["rows"])).hex(),
                  "requests": res})


def grid_nev_fixture():
    """SURVEY 8(d)'s n_evidence variants of the C3 stream on the 10x10 K=4 grid: for n_evidence in {1, 8, 16} the first requests
    of netspec.c3_requests(..., n_evidence, seed=1) whose row-major product (the order the hash-ordered reference eliminates in)
    stays below 3e6 rows - seconds each in the reference.  -> grid10x10_nev.json"""
    spec = netspec.grid_spec(10, 10, 4, seed=0)
    bn = netspec.build(spec, sorobn.BayesNet, wrap=refload.HashedName)
    out = {"recipe": {"R": 10, "C": 10, "K": 4, "seed": 0},
           "cpt_sum_hex": float(sum(r[-1] for c in spec["cpts"].values() for r in c["rows"])).hex(), "variants": {}}
    for n_ev in (1, 8, 16):
        q, ev, ec = netspec.c3_requests(100, 4, 2000, n_ev, seed=1)
        reqs, stream_index = [], []
        for i in range(2000):
            if netspec.grid_row_major_cost(q[i], ev[i], 10, 10, 4)[0] <= 3e6:
                reqs.append(((f"{q[i]:03d}",), [(f"{e:03d}", int(c)) for e, c in zip(ev[i], ec[i])]))
                stream_index.append(i)
            if len(reqs) >= 8:
                break
        t0 = time.time()
        res = run_requests(bn, reqs, refload.HashedName)
        for r, i in zip(res, stream_index):
            r["stream_index"] = i
        out["variants"][str(n_ev)] = res
        print("grid10x10 n_evidence", n_ev, len(reqs), round(time.time() - t0, 1), "s", [r["ref_seconds"] for r in res], flush=True)
    netspec.save(os.path.join(HERE, "grid10x10_nev.json"), out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--heavy", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["examples", "impute", "dags", "grids", "wide", "many", "huge", "joint", "learn"]
    if "examples" in todo:
        examples_fixture()
    if "impute" in todo:
        impute_fixture()
    if "dags" in todo:
        dags_fixture()
    if "grids" in todo:
        grids_fixture(a.heavy)
    if "nev" in todo or "grids" in todo:
        grid_nev_fixture()
    if "wide" in todo:
        wide_fixture()
    if "many" in todo:
        many_nodes_fixture()
    if "huge" in todo:
        huge_fixture()
    if "joint" in todo:
        joint_fixture()
    if "learn" in todo:
        learn_fixture()
