"""One-hot (log2) observation encoding.  Mirror of ``twentyfortyeight/utils/binary.py``."""

from __future__ import annotations

import numpy as np
from numpy import int64, ndarray

from ... import ops, scalar


def encode(state: ndarray, encodage_size: int) -> ndarray:
    """[ref binary.py:11-50] eye(encodage_size)[log2(state)] (0 stays index 0): shape (state.size, size)."""
    flat = np.asarray(state).reshape(-1).astype(np.int64)
    cells = flat.size
    sc = scalar.ScalarOps
    if cells <= sc.ONEHOT_MAX_CELLS and 1 <= int(encodage_size) <= sc.ONEHOT_MAX_WIDTH:  # one board: the pinned-operand path
        return scalar.current().onehot(flat, int(encodage_size)).astype(int64)
    if cells % 16:  # the kernel encodes 16 cells per record: pad (boards of other sizes than 4 x 4), trim below
        flat = np.concatenate([flat, np.zeros(-cells % 16, dtype=np.int64)])
    packed = ops.pack(flat.reshape(-1, 16))
    out = ops.onehot(packed, int(encodage_size)).cpu().numpy()
    return out.reshape(flat.size, int(encodage_size))[:cells].astype(int64)


def encode_flatten(state: ndarray, encodage_size: int) -> ndarray:
    """[ref binary.py:53-86] flatten, then encode, then ravel: length state.size * encodage_size."""
    return encode(np.asarray(state).ravel(), encodage_size).ravel()
