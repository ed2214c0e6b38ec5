"""TEST DOUBLE for ``jax.random`` (see jax/__init__.py): keys are (2,) uint32 NumPy arrays; split derives children with a
NumPy generator seeded by the parent -- deterministic, distinct, and NOT threefry (no test depends on the bitstream)."""

import numpy as _np


def PRNGKey(seed):
    return _np.array([0, int(seed) & 0xFFFFFFFF], dtype=_np.uint32)


def split(key, num=2):
    k = _np.asarray(key).astype(_np.uint64)
    rng = _np.random.default_rng([int(k[0]), int(k[1]), 0x6A09E667])
    return rng.integers(0, 1 << 32, size=(num, 2), dtype=_np.uint64).astype(_np.uint32)
