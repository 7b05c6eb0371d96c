"""Helpers to read golden vectors (tests/golden/*.json) and compare answers against them."""
import os

import numpy as np

import netspec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9  # north_star: marginals within 1e-9 of the pandas reference; index exact


def load(name):
    return netspec.load(os.path.join(GOLDEN, name))


def expected(req):
    e = req["expect"]
    vals = np.array([float.fromhex(h) for h in e["values_hex"]], dtype=np.float64)
    rows = [tuple(r) for r in e["index"]]
    return e["name"], e["index_names"], rows, vals, e["multi"]


def same_label(a, b):
    # bool vs int must not be confused in an *index* comparison
    return type(a) is type(b) and a == b


def assert_rows_equal(got_rows, exp_rows, ctx=""):
    assert len(got_rows) == len(exp_rows), f"{ctx}: {len(got_rows)} rows, expected {len(exp_rows)}"
    for g, e in zip(got_rows, exp_rows):
        g = g if isinstance(g, tuple) else (g,)
        assert len(g) == len(e) and all(same_label(netspec._py(x), y) for x, y in zip(g, e)), \
            f"{ctx}: index row {g} != {e}"


def grid_spec_from_recipe(entry):
    r = entry["recipe"]
    spec = netspec.grid_spec(r["R"], r["C"], r["K"], seed=r["seed"])
    s = float(sum(row[-1] for c in spec["cpts"].values() for row in c["rows"]))
    assert s.hex() == entry["cpt_sum_hex"], "grid recipe drifted from the golden generator"
    return spec


def dag_spec_from_recipe(entry):
    """huge_cards.json stores the recipe of a random DAG (cardinalities up to 100: its CPTs would be megabytes of JSON), not the
    CPTs; the sum of all CPT numbers pins the regenerated network to the one the reference answered."""
    r = dict(entry["recipe"])
    name = r.pop("name")
    spec = netspec.random_dag_spec(r.pop("seed"), cards=tuple(r.pop("cards")), **r)
    spec["name"] = name
    s = float(sum(row[-1] for c in spec["cpts"].values() for row in c["rows"]))
    assert s.hex() == entry["cpt_sum_hex"], "DAG recipe drifted from the golden generator"
    return spec
