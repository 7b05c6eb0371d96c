"""TEST INFRASTRUCTURE ONLY - loader for the *unmodified* reference implementation.

Imports MaxHalford/sorobn from /root/reference (read-only, only present in the build container) so that
golden vectors can be generated from it - or, where /root/reference does not exist (the GPU box), from
`oracle/_ref/`: the same modules byte-compiled by `oracle/build_ref.py` (`make -C oracle _ref`), which
travel with the push like the in-tree `.so` files.  bench.py's cpu_baseline leg and the `-m gpu` drop-in
tests run the reference's own pandas path from there on the GPU box's host cores.

The reference does `import vose` (bayes_net.py:10), a third-party Cython package
(vose 0.2.5, uv.lock:506-512) that is not installed and not installable here.  The exact
path never calls it (only bayes_net.py:37-41 do), so a small numpy stand-in is registered
in sys.modules before the import.  The stand-in is NOT the real alias sampler: anything that
depends on the random stream (sample / gibbs / likelihood / rejection) is "parity unpinned".

Elimination order: bayes_net.py:766,779 iterate a Python set.  For str names that order
depends on PYTHONHASHSEED; `HashedName` is a str subclass whose hash is its integer value, so a
set of zero-padded ids iterates in ascending order (CPython small-int hashing), which forces the
unmodified reference into ascending-id elimination.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
REF_BUILD_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


class HashedName(str):
    """str subclass hashing to int(self): makes the reference eliminate in ascending-id order."""

    __slots__ = ()

    def __hash__(self):
        return int(self)

    __eq__ = str.__eq__


def _install_vose_stub():
    if "vose" in sys.modules:
        return
    import numpy as np

    class Sampler:  # signature of vose.Sampler(weights=float64[], seed=int)
        def __init__(self, weights, seed=None):
            w = np.asarray(weights, dtype=float)
            self._cdf = np.cumsum(w / w.sum())
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(min(np.searchsorted(self._cdf, self._rng.random(), side="right"),
                           len(self._cdf) - 1))

    m = types.ModuleType("vose")
    m.Sampler = Sampler
    sys.modules["vose"] = m


def source_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sorobn"))


def build_available() -> bool:
    """oracle/_ref/ holds the byte-compiled reference and its bytecode matches this interpreter."""
    import importlib.util
    import json

    try:
        with open(os.path.join(REF_BUILD_ROOT, "MANIFEST.json")) as f:
            magic = json.load(f)["magic"]
    except (OSError, ValueError, KeyError):
        return False
    return magic == importlib.util.MAGIC_NUMBER.hex() and os.path.exists(os.path.join(REF_BUILD_ROOT, "sorobn", "bayes_net.pyc"))


def available() -> bool:
    return source_available() or build_available()


def load(which="auto"):
    """Return the reference `sorobn` module (unmodified).  which: "source" (/root/reference), "build"
    (oracle/_ref, sourceless bytecode of the same files) or "auto" (source where mounted, else build)."""
    if "sorobn" in sys.modules and getattr(sys.modules["sorobn"], "_mibn_refload", None):
        return sys.modules["sorobn"]
    if which == "auto":
        which = os.environ.get("MIBN_REFLOAD", "auto")  # "build": behave like the GPU box, where only oracle/_ref exists
    if which == "auto":
        which = "source" if source_available() else "build"
    if which == "source" and not source_available():
        raise RuntimeError("reference not mounted at /root/reference")
    if which == "build" and not build_available():
        raise RuntimeError("oracle/_ref is missing or was compiled by another Python: run `make -C oracle _ref` "
                           "where /root/reference is mounted")
    _install_vose_stub()
    root = REFERENCE_ROOT if which == "source" else REF_BUILD_ROOT
    if root not in sys.path:
        sys.path.insert(0, root)
    import sorobn  # noqa: E402

    sorobn._mibn_refload = which
    return sorobn
