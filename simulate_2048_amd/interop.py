"""
Array interchange at the edge of the API mirrors, so that the reference's callers run UNCHANGED on top of them
(reinforce/training/self_play.py:298-300,316-317,346-356,374-375 and reinforce/evaluate.py:89-128):

* what the mirrors RETURN is a ``DeviceArray`` -- a ``torch.Tensor`` (same storage, still on the GPU, every torch
  operation works and yields DeviceArrays again; ``.dtype`` is the real ``torch.dtype``, so ``torch.zeros(n, dtype=x.dtype)``,
  ``y.to(x.dtype)``, ``torch.finfo(x.dtype)`` and dtype-keyed dicts behave as for any tensor -- NumPy-side questions are
  answered by ``np.array(x).dtype``) that NumPy can swallow: ``np.array(x)`` / ``np.asarray(x)`` make ONE
  device-to-host copy, which is what the caller's ``np.array(observations)`` etc. mean on a JAX array too.
  ``__dlpack__`` (inherited from torch) hands the device buffer to any DLPack consumer (JAX on ROCm, CuPy, torch)
  without a copy; ``bool() / int() / float()`` read a scalar back as on a JAX array.
* what the mirrors ACCEPT is anything array-like: torch tensors (any device), NumPy arrays, Python scalars/sequences
  and any ``__dlpack__`` / ``__array__`` producer (JAX arrays: ``(N, 2)`` uint32 PRNG keys, ``(N,)`` int32 actions).
"""

from __future__ import annotations

import numpy as np
import torch


# id(tensor) -> (ResultSet, field name, the tensor itself, un-batched alias?) for the tensors of pooled result sets; resultpool
# (which imports this module for DeviceArray) fills it.  The third entry pins the identity: an id can be reused by another
# object once the original is gone, the stored reference cannot be anything else.
_registry: dict = {}
_registry_get = _registry.get


class DeviceArray(torch.Tensor):
    """A torch tensor (usually on the GPU) that converts to NumPy on demand -- see the module docstring."""

    def __deepcopy__(self, memo):
        return self.as_subclass(torch.Tensor).clone().as_subclass(DeviceArray)

    def __reduce_ex__(self, protocol):  # pickles as the plain tensor it is
        return self.as_subclass(torch.Tensor).__reduce_ex__(protocol)

    @staticmethod
    def wrap(t: torch.Tensor) -> 'DeviceArray':
        return t if isinstance(t, DeviceArray) else t.as_subclass(DeviceArray)

    def __array__(self, dtype=None, copy=None):
        # a field of a pooled result set of the reinforce.game mirrors: one wait per LAUNCH, then host-side copies
        hit = _registry_get(id(self))
        if hit is not None and hit[2] is self:
            host = hit[0].host_array(hit[1])
            if host is not None:
                if hit[3]:  # the un-batched alias of a one-environment set: row 0
                    host = host[0]
                return host.astype(dtype) if dtype is not None and dtype != host.dtype else np.array(host)
        plain = self.as_subclass(torch.Tensor).detach()
        if plain.dtype == torch.bfloat16:  # NumPy has no bfloat16: widen (exact)
            plain = plain.to(torch.float32)
        host = plain.cpu().numpy()
        if dtype is not None:
            host = host.astype(dtype, copy=False)
        return host.copy() if copy and plain.device.type == 'cpu' else host

    def numpy(self, *args, **kwargs):  # JAX-style / torch-style explicit conversion, also from the GPU
        return self.__array__()

    # Scalar read-backs (`bool(done)`, `float(reward)`, `int(info['max_tile'])` of the reference's single-game loops,
    # reinforce/training/self_play.py:102-135, evaluate.py:100-128): through __array__, i.e. from the host mirror of a pooled
    # result set when there is one (no device-to-host copy, no stream synchronisation), else torch's own .item().
    def item(self):
        hit = _registry_get(id(self))
        if hit is not None and hit[2] is self:
            return self.__array__().item()
        return self.as_subclass(torch.Tensor).item()

    def __bool__(self):
        return bool(self.item())

    def __float__(self):
        return float(self.item())

    def __int__(self):
        return int(self.item())

    def __index__(self):  # as torch: only integer (and bool) scalars index or count; a float reward must not truncate silently
        if self.dtype.is_floating_point or self.dtype.is_complex:
            raise TypeError(f'only integer tensors of a single element can be converted to an index (got {self.dtype})')
        return int(self.item())


def wrap(t):
    """DeviceArray view of a tensor (None stays None)."""
    return None if t is None else t.as_subclass(DeviceArray)


def to_tensor(x) -> torch.Tensor:
    """Anything array-like -> torch.Tensor WITHOUT changing device (a later ``.to(device)`` moves it): torch tensors as
    they are, DLPack producers zero-copy when torch can import them, everything else through NumPy."""
    if isinstance(x, torch.Tensor):
        # internal arithmetic runs on the plain tensor: a DeviceArray would send every small op through
        # __torch_function__ (measured: +8 us per mirror call at N = 8)
        return x.as_subclass(torch.Tensor) if type(x) is not torch.Tensor else x
    if not isinstance(x, (np.ndarray, np.generic, int, float, bool, list, tuple)) and hasattr(x, '__dlpack__'):
        try:
            return torch.from_dlpack(x)
        except Exception:  # a producer torch cannot import (other device type, unsupported dtype): host copy instead
            pass
    a = np.asarray(x)
    if a.dtype == np.uint32:  # same bits as int32 (keys); torch's uint32 support is partial
        a = np.ascontiguousarray(a).view(np.int32)
    elif a.dtype == np.uint64:
        a = np.ascontiguousarray(a).view(np.int64)
    elif a.dtype == np.uint16:
        a = a.astype(np.int32)
    return torch.from_numpy(np.ascontiguousarray(a))


def key_words(key) -> tuple[int, int]:
    """One PRNG key -> its two 32-bit words (k0, k1): a (2,) uint32 array-like (JAX PRNGKey / threefry key data), or a
    plain integer seed (k0 = high word, k1 = low word)."""
    if isinstance(key, (int, np.integer)):
        v = int(key) & 0xFFFFFFFFFFFFFFFF
        return v >> 32, v & 0xFFFFFFFF
    flat = to_tensor(key).reshape(-1).cpu().tolist()
    if len(flat) == 1:
        return key_words(int(flat[0]))
    return int(flat[0]) & 0xFFFFFFFF, int(flat[1]) & 0xFFFFFFFF


def keys_on_device(keys, device) -> torch.Tensor:
    """(N, 2) PRNG keys -> contiguous int32 (N, 2) tensor on `device` holding the uint32 bit patterns."""
    t = to_tensor(keys)
    if t.dim() == 1 and t.numel() == 2:
        t = t.reshape(1, 2)
    if t.dim() != 2 or t.shape[1] != 2:
        raise ValueError(f'expected PRNG keys of shape (N, 2), got {tuple(t.shape)}')
    if t.dtype == torch.int32:
        pass
    elif t.dtype in (torch.int64, torch.uint8, torch.int16, torch.int8):
        t = t.to(torch.int64).bitwise_and(0xFFFFFFFF)
        t = torch.where(t >= 1 << 31, t - (1 << 32), t).to(torch.int32)  # keep the low 32 bits as a bit pattern
    elif hasattr(torch, 'uint32') and t.dtype == torch.uint32:
        t = t.view(torch.int32)
    else:
        raise TypeError(f'PRNG keys must be integers, got {t.dtype}')
    return t.to(device).contiguous()
