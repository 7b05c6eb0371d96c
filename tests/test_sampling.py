"""SURVEY.md section 8f rank 2 - sample / rejection sampling / likelihood weighting (bayes_net.py:518-663).
Parity with the reference's random stream is unpinned (it needs the absent `vose` sampler), so the GPU kernels are
checked statistically: against the exact posterior (sample marginals, rejection sampling) and against the exact
limit of the reference's likelihood-weighting estimator, computed by enumeration.  A CPU-only test confirms that
limit against the unmodified reference itself (build container only)."""
import itertools
import os

import numpy as np
import pandas as pd
import pytest

import golden_util as gu
import netspec
import sorobn_amd
from sorobn_amd.flatten import flatten


def _spec(name):
    return next(n for n in gu.load("examples.json") if n["spec"]["name"] == name)["spec"]


def llh_weighting_limit(bn, query, event):
    """What _llh_weighting (bayes_net.py:621-663) converges to: samples are drawn with the event clamped
    (g = product of P(v | pa) over the free variables), each carries likelihood L = product of P(v | pa) over ALL
    variables, the answer is mean(L | query state), normalised."""
    f = flatten(bn)
    n = len(f.names)
    ev = {f.id[k]: f.code_of(f.id[k], v) for k, v in event.items()}
    num, den = {}, {}
    for s in itertools.product(*[range(int(c)) for c in f.card]):
        if any(s[v] != c for v, c in ev.items()):
            continue
        g = L = 1.0
        for v in range(n):
            sc = f.scope[v]
            off = 0
            for u in sc:
                off = off * int(f.card[u]) + s[u]
            p = f.values[f.value_off[v] + off]
            L *= p
            if v not in ev:
                g *= p
        key = tuple(s[f.id[q]] for q in query)
        num[key] = num.get(key, 0.0) + g * L
        den[key] = den.get(key, 0.0) + g
    mean = {k: num[k] / den[k] for k in num if den[k] > 0}
    z = sum(mean.values())
    return {k: v / z for k, v in mean.items()}


@pytest.mark.skipif(not os.path.isdir("/root/reference/sorobn"), reason="reference only exists in the build container")
def test_reference_likelihood_weighting_converges_to_the_enumerated_limit():
    from oracle import refload
    ref = refload.load().examples.sprinkler()
    ref._rng.seed(5)
    got = ref.query("Rain", event={"Sprinkler": True}, algorithm="likelihood", n_iterations=6000)
    mine = netspec.build(_spec("sprinkler"), sorobn_amd.BayesNet)
    lim = llh_weighting_limit(mine, ("Rain",), {"Sprinkler": True})
    f = flatten(mine)
    for lab, p in got.items():
        assert abs(p - lim[(f.code_of(f.id["Rain"], lab),)]) < 0.03


@pytest.mark.gpu
def test_forward_samples(amd=None):
    bn = netspec.build(_spec("asia"), sorobn_amd.BayesNet)
    one = bn.sample()
    assert isinstance(one, pd.Series) and list(one.index) == list(bn.nodes)
    df = bn.sample(200_000)
    assert list(df.columns) == sorted(bn.nodes) and len(df) == 200_000
    for node in bn.nodes:  # marginals of the samples vs the exact marginals
        exact = bn.query(node, event={})
        freq = df[node].value_counts(normalize=True)
        for lab, p in exact.items():
            assert abs(freq.get(lab, 0.0) - p) < 0.005, (node, lab)
    forced = bn.sample(1000, init={"Smoker": True, "Visit to Asia": True})
    assert forced["Smoker"].all() and forced["Visit to Asia"].all()
    # joint structure: P(Dispnea | Bronchitis) from the samples vs the CPT-implied exact conditional
    exact = bn.query("Dispnea", event={"Bronchitis": True})
    sel = df[df["Bronchitis"]]
    assert abs(sel["Dispnea"].mean() - exact[True]) < 0.01
    assert not bn.sample(50).equals(bn.sample(50))  # successive calls advance the stream
    with pytest.raises(ValueError, match="Unknown method"):
        bn.sample(2, method="nope")


@pytest.mark.gpu
def test_rejection_sampling_matches_exact():
    bn = netspec.build(_spec("sprinkler"), sorobn_amd.BayesNet)
    ans = bn.query("Rain", event={"Sprinkler": True}, algorithm="rejection", n_iterations=2_000_000)
    assert ans.name == "P(Rain)" and ans.index.name == "Rain" and ans.index.tolist() == [False, True]
    assert abs(ans.sum() - 1.0) < 1e-12
    assert abs(ans[False] - 0.7) < 0.005  # bayes_net.py:751-755: exact 0.7 / 0.3
    two = bn.query("Rain", "Cloudy", event={"Wet grass": True}, algorithm="rejection", n_iterations=2_000_000)
    exact = bn.query("Rain", "Cloudy", event={"Wet grass": True})
    assert two.name == "P(Rain, Cloudy)" and list(two.index.names) == ["Cloudy", "Rain"]
    assert float(np.max(np.abs(two.reindex(exact.index).to_numpy() - exact.to_numpy()))) < 0.005
    empty = bn.query("Rain", event={"Sprinkler": "no-such-label"}, algorithm="rejection", n_iterations=1000)
    assert len(empty) == 0 and empty.name == "P(Rain)"


@pytest.mark.gpu
def test_likelihood_weighting_matches_the_reference_estimator():
    for name, q, ev in [("sprinkler", ("Rain",), {"Sprinkler": True}),
                        ("asia", ("Lung cancer", "Bronchitis"), {"Smoker": True, "Dispnea": False}),
                        ("grades", ("Intelligence",), {"Letter": "Strong"})]:
        spec = _spec(name)
        bn = netspec.build(spec, sorobn_amd.BayesNet)
        if name == "grades":  # pick an existing label of the evidence variable
            lab = flatten(bn).domains[flatten(bn).id["Letter"]][0]
            ev = {"Letter": lab}
        ans = bn.query(*q, event=ev, algorithm="likelihood", n_iterations=2_000_000)
        lim = llh_weighting_limit(bn, q, ev)
        f = flatten(bn)
        assert abs(ans.sum() - 1.0) < 1e-12
        order = sorted(q)
        for key, p in ans.items():
            key = key if isinstance(key, tuple) else (key,)
            codes = {n: f.code_of(f.id[n], lab) for n, lab in zip(order, key)}
            assert abs(p - lim[tuple(codes[n] for n in q)]) < 0.005, (name, key)


@pytest.mark.gpu
def test_sampling_walk_vs_the_references_own_arithmetic():
    """f2's deterministic half (VERDICT r3 missing #4): everything in the sampling paths that is a pure function of the CPTs,
    computed by sample_kernel ITSELF (mibn_sample_probe: its walk over given joint states) against the reference's arithmetic
    on live reference objects (oracle/_ref):
      * the likelihood weight of a sample - `_forward_sample(init=sample)` clamps every node and therefore draws nothing; the
        generator's second value is the product _llh_weighting averages (bayes_net.py:541-546, 646-652)           <= 1e-12 relative
      * the conditional row every node is drawn from - `P.cdt[condition]` (44-52, 530-534), the weights the reference hands to
        its alias sampler (28-42), as cumulative probabilities                                                       <= 1e-12
    Only the random stream itself stays unpinned (the reference's needs the absent `vose`)."""
    from oracle import refload
    if not refload.available():
        pytest.skip("oracle/_ref was not built (make -C oracle _ref where /root/reference is mounted)")
    ref_mod = refload.load()
    n_checked = 0
    for name in ("alarm", "sprinkler", "asia", "grades"):
        ref = getattr(ref_mod.examples, name)()
        bn = netspec.build(_spec(name), sorobn_amd.BayesNet)
        f = flatten(bn)
        eng = bn.backend.engine
        assert list(ref.nodes) == list(f.names)
        states = eng.sample(96, seed=11)  # states of positive probability: every conditional row exists in the sparse CPTs
        lik, cdf = eng.sample_probe(states)
        for r, st in enumerate(states):
            sample = {f.names[v]: f.domains[v][int(st[v])] for v in range(len(f.names))}
            got_sample, want_lik = next(ref._forward_sample(init=sample))
            assert got_sample == sample
            assert want_lik > 0 and abs(lik[r] - want_lik) <= 1e-12 * want_lik, (name, r, lik[r], want_lik)
            for v, node in enumerate(f.names):
                P = ref.P[node]
                if node in ref.parents:
                    P = P.cdt[tuple(sample[p] for p in ref.parents[node])]
                w = np.array([float(P.get(lab, 0)) for lab in f.domains[v]])
                assert abs(w.sum() - float(P.to_numpy(dtype=float).sum())) <= 1e-15  # (no row outside the flattened domain)
                k = int(f.card[v])
                got = cdf[r, v, :k]
                assert abs(got[-1] - w.sum()) <= 1e-12
                assert np.max(np.abs(got / got[-1] - np.cumsum(w) / w.sum())) <= 1e-12, (name, node)
                n_checked += 1
    assert n_checked >= 96 * (5 + 4 + 8 + 5)
    # a state of probability zero under a sparse CPT: absent rows are zeros of the dense table, P.get(value, 0) of the reference
    bn = netspec.build(_spec("asia"), sorobn_amd.BayesNet)
    f = flatten(bn)
    st = np.zeros((1, len(f.names)), np.uint8)
    for v, nme in enumerate(f.names):
        st[0, v] = f.code_of(v, {"TB or cancer": False, "Tuberculosis": True}.get(nme, False))
    lik, _ = bn.backend.engine.sample_probe(st)
    assert lik[0] == 0.0
