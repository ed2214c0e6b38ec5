"""TEST DOUBLE for ``jax.numpy`` (see jax/__init__.py): NumPy functions that first convert their arguments with np.asarray,
so they also accept the mirrors' DeviceArrays (device -> host copy)."""

import numpy as _np



class _ScalarType:
    """Stand-in for a ``jnp`` scalar type (``jnp.float32`` ...): callable and usable as ``dtype=`` like the NumPy type it
    wraps, and EQUAL to the torch dtype of the same element type -- the arrays under test are torch tensors whose ``.dtype``
    is a real ``torch.dtype``, and the reference's test file writes ``x.dtype == jnp.float32``.  The vocabulary bridge belongs
    here, in the test double, not in the product's return type."""

    def __init__(self, np_type):
        self._np = np_type
        self.dtype = _np.dtype(np_type)  # NumPy's protocol: np.dtype(obj) takes obj.dtype

    def __call__(self, x=0):
        return self._np(_h(x))

    def __eq__(self, other):
        if type(other).__module__ == 'torch' and type(other).__name__ == 'dtype':
            return str(other) == 'torch.' + self.dtype.name
        if isinstance(other, _ScalarType):
            return self.dtype == other.dtype
        try:
            return self.dtype == _np.dtype(other)
        except TypeError:
            return NotImplemented

    def __ne__(self, other):
        result = self.__eq__(other)
        return result if result is NotImplemented else not result

    def __hash__(self):
        return hash(self.dtype)

    def __repr__(self):
        return f'jnp.{self.dtype.name}'


int32, int64, float32, bool_, uint32, uint8 = (_ScalarType(t) for t in (_np.int32, _np.int64, _np.float32, _np.bool_, _np.uint32, _np.uint8))
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
