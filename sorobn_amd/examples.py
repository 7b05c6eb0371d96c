"""The four example networks of the reference (`sorobn.examples.alarm / asia / grades / sprinkler`, sorobn/examples.py:8-324)
as `sorobn_amd.BayesNet` objects, so that code written against `sorobn.examples.asia()` ports by changing the import.

The structures and CPT numbers are data: they are read from `data/example_networks.json`, the network specs
`tests/golden/make_golden.py` dumped from the reference's own objects (the same specs the golden tests use), not from
the reference's source.  Label types (bool / str) are preserved; `prepare()` stores every CPT with its rows sorted (the reference's `alarm()` leaves
multi-parent CPTs in the order they were typed - the posteriors do not depend on it).
"""
import json
import os

import pandas as pd

from .bayes_net import BayesNet

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "example_networks.json")
_specs = None


def _build(name):
    global _specs
    if _specs is None:
        with open(_DATA) as f:
            _specs = json.load(f)
    spec = _specs[name]
    in_edges = {n for e in spec["edges"] for n in e}
    bn = BayesNet(*[(p, c) for p, c in spec["edges"]], *[n for n in spec["nodes"] if n not in in_edges])
    for node, cpt in spec["cpts"].items():
        rows = cpt["rows"]
        if len(cpt["names"]) == 1:
            idx = pd.Index([r[0] for r in rows], name=cpt["names"][0])
        else:
            idx = pd.MultiIndex.from_tuples([tuple(r[:-1]) for r in rows], names=cpt["names"])
        bn.P[node] = pd.Series([r[-1] for r in rows], index=idx)
    bn.prepare()
    return bn


def alarm():
    """Pearl's burglary alarm network (examples.py:8-80): 5 boolean nodes."""
    return _build("alarm")


def asia():
    """Lauritzen & Spiegelhalter's Asia network (examples.py:83-176): 8 boolean nodes."""
    return _build("asia")


def grades():
    """Koller & Friedman's student network (examples.py:241-324): 5 nodes, one with three states."""
    return _build("grades")


def sprinkler():
    """The sprinkler network (examples.py:179-238): 4 boolean nodes."""
    return _build("sprinkler")
