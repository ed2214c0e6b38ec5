"""
TEST DOUBLE, NOT JAX.  Just enough of the ``jax`` / ``jax.numpy`` / ``jax.random`` surface (NumPy underneath) for the
reference's tests/test_jax_game.py to be COLLECTED AND RUN AGAINST THIS REPO'S ``reinforce.game`` MIRRORS in the build
container, where the real JAX is not installed (and cannot be: no network).  Nothing of the reference's engine runs
through it (the aliased ``reinforce.game`` modules are simulate_2048_amd's), the product never imports it, and it is only
put on the path by tests/test_reference_suite_on_mirrors.py for that one subprocess.
jit = identity; vmap = a Python loop over the leading axis that stacks the results.
"""

import numpy as _np

from . import numpy, random  # noqa: F401  (submodules: jax.numpy, jax.random)


def jit(fn=None, **_kwargs):
    if fn is None:
        return lambda f: f
    return fn


def _stack(items):
    first = items[0]
    if isinstance(first, tuple):
        return tuple(_stack([it[k] for it in items]) for k in range(len(first)))
    return _np.stack([_np.asarray(it) for it in items])


def vmap(fn, in_axes=0, out_axes=0):
    def batched(*args):
        n = len(_np.asarray(args[0]))
        return _stack([fn(*[a[i] for a in args]) for i in range(n)])

    return batched
