"""TEST DOUBLE for ``jax.numpy`` (see jax/__init__.py): NumPy functions that first convert their arguments with np.asarray,
so they also accept the mirrors' DeviceArrays (device -> host copy)."""

import numpy as _np

int32, int64, float32, bool_, uint32, uint8 = _np.int32, _np.int64, _np.float32, _np.bool_, _np.uint32, _np.uint8
ndarray = _np.ndarray


def _h(x):
    return _np.asarray(x)


def array(x, dtype=None):
    return _np.array(_h(x), dtype=dtype)


asarray = array


def zeros(shape, dtype=float32):
    return _np.zeros(shape, dtype=dtype)


def ones(shape, dtype=float32):
    return _np.ones(shape, dtype=dtype)


def array_equal(a, b):
    return bool(_np.array_equal(_h(a), _h(b)))


def sum(x, axis=None):  # noqa: A001
    return _np.sum(_h(x), axis=axis)


def isclose(a, b, rtol=1e-5, atol=1e-8):
    return _np.isclose(_h(a), _h(b), rtol=rtol, atol=atol)


def allclose(a, b, rtol=1e-5, atol=1e-8):
    return bool(_np.allclose(_h(a), _h(b), rtol=rtol, atol=atol))


def all(x, axis=None):  # noqa: A001
    return _np.all(_h(x), axis=axis)


def any(x, axis=None):  # noqa: A001
    return _np.any(_h(x), axis=axis)


def max(x, axis=None):  # noqa: A001
    return _np.max(_h(x), axis=axis)


def log2(x):
    return _np.log2(_h(x))


def where(c, a, b):
    return _np.where(_h(c), _h(a), _h(b))
