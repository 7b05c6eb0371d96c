"""sorobn_amd - MI355X-native exact-inference backend for sorobn-style Bayesian networks.

Only the hot path of MaxHalford/sorobn is implemented:
`BayesNet.query(algorithm="exact" | "gibbs")` and `BayesNet.impute`, with the reference's API and
pandas return types, executed by hand-written gfx950 HIP kernels behind the C-ABI in
include/mibn.h (libmibn.so, loaded through ctypes - no PyTorch on the product path).
"""
from .bayes_net import Backend, BayesNet, accelerate

__all__ = ["BayesNet", "Backend", "accelerate"]
__version__ = "0.1.0"
