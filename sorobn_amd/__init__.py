"""sorobn_amd - MI355X-native exact-inference backend for sorobn-style Bayesian networks.

The hot path of MaxHalford/sorobn - `BayesNet.query(algorithm="exact" | "gibbs")` and
`BayesNet.impute` - and the "next" rows of section 8f (full_joint_dist / predict_proba, sample / rejection / likelihood
weighting, fit / partial_fit, structure.chow_liu), with the reference's API and pandas return types, executed by
hand-written gfx950 HIP kernels behind the C-ABI in include/mibn.h (libmibn.so, loaded through ctypes - no PyTorch on
the product path).
"""
from . import examples, structure
from .bayes_net import Backend, BayesNet, accelerate

__all__ = ["BayesNet", "Backend", "accelerate", "examples", "structure"]
__version__ = "0.1.0"
