"""TEST INFRASTRUCTURE ONLY - loader for the *unmodified* reference implementation.

Imports MaxHalford/sorobn from /root/reference (read-only, only present in the build
container, never on the GPU box) so that golden vectors can be generated from it.

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


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sorobn"))


def load():
    """Return the reference `sorobn` module (unmodified), or raise if it is not mounted."""
    if not available():
        raise RuntimeError("reference not mounted at /root/reference (expected on the GPU box)")
    _install_vose_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import sorobn  # noqa: E402

    return sorobn
