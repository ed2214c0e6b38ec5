"""
Process-wide runtime used by the drop-in API mirrors: which C-ABI library and which device the
single-board / functional shims run on.

Default = the HIP library on the current HIP device; creating it without a usable GPU raises (no
CPU fallback).  ``use(lib, device)`` swaps it temporarily -- the test-suite uses that to run the
host-side logic of the shims against an oracle-backed stand-in on CPU tensors.
"""

from __future__ import annotations

import contextlib
import os
import threading

import torch

from . import capi


# torch's current stream of a device as a raw hipStream_t handle: the private accessor inductor uses costs 0.2 us, the public
# Stream object 2.2 us -- a sixth of the host work of a reference-sized move.  Falls back to the public API when it is gone.
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class Runtime:
    def __init__(self, lib, device):
        self.lib = lib
        self.device = torch.device(device)
        self._index = self.device.index if self.device.type == 'cuda' else -1

    def handles(self):
        """(device ordinal, hipStream_t of torch's current stream) for a C-ABI call."""
        if self.device.type != 'cuda':
            return -1, None
        index = self._index if self._index is not None else torch.cuda.current_device()
        if _raw_stream is not None:
            return index, _raw_stream(index) or None
        return index, torch.cuda.current_stream(self.device).cuda_stream or None

    def synchronize(self) -> None:
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)


_current: Runtime | None = None


def current() -> Runtime:
    global _current
    if _current is None:
        lib = capi.library()
        if lib.device_count() <= 0 or not torch.cuda.is_available():
            raise RuntimeError('simulate_2048_amd needs a HIP device: there is no CPU fallback')
        _current = Runtime(lib, torch.device('cuda', torch.cuda.current_device()))
    return _current


@contextlib.contextmanager
def use(lib, device):
    """Temporarily run the API mirrors on another library/device pair."""
    global _current
    saved = _current
    _current = Runtime(lib, device)
    try:
        yield _current
    finally:
        _current = saved


# Unseeded calls of the reference draw from a module-level OS-entropy generator
# (twentyfortyeight/core/gameboard.py:20) and are not reproducible; here they walk a Philox stream
# keyed by OS entropy.
_entropy_seed = int.from_bytes(os.urandom(8), 'little')
_entropy_lock = threading.Lock()
_entropy_next = 0


def entropy_stream(count: int = 1) -> tuple[int, int]:
    """(seed, first step_index) for one unseeded call that consumes ``count`` consecutive step words
    (``fill_cells(state, number_tile=k)`` spawns with steps c .. c+k-1): the counter advances by ``count``,
    so no two unseeded calls ever share a (seed, step_index, env) word -- like the reference's continuing
    module-level generator (gameboard.py:20), every draw is fresh."""
    global _entropy_next
    count = max(1, int(count))
    with _entropy_lock:
        first = _entropy_next
        _entropy_next += count
    return _entropy_seed, first


def seed_to_u64(seed) -> int:
    """Reference seeds (Python ints for NumPy, 2 x uint32 PRNGKeys for JAX -- any array-like) -> 64-bit Philox key
    (k0 << 32) | k1, the same mapping the keyed C-ABI entry points apply to a key row."""
    from .interop import key_words

    k0, k1 = key_words(seed)
    return (k0 << 32) | k1
