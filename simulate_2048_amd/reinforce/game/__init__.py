"""
Drop-in mirror of ``reinforce/game`` (the JAX restatement of the 2048 engine and its functional env
API): same function names, argument meaning and result shapes/dtypes, with ``torch`` tensors on the
GPU standing in for JAX arrays and every board computation done by the HIP kernels.

PRNG keys: any ``(2,)`` uint32 array (a JAX ``PRNGKey``) or Python int is folded into the 64-bit
Philox key; batched calls take ``(N, 2)`` keys and fold the FIRST one, environment i then draws
from the global-index stream i (include/env2048.h, RNG contract).
"""
