"""Shared helpers of the NumPy-facing mirrors: value arrays <-> packed device boards."""

from __future__ import annotations

import numpy as np
import torch

from ... import ops, runtime


def require_4x4(board: np.ndarray, what: str) -> None:
    if board.ndim != 2 or board.shape != (4, 4):
        raise ValueError(
            f'{what}: the MI355X engine handles 4x4 boards only (SURVEY.md App. A.5), got shape {board.shape}'
        )


def to_device(board) -> torch.Tensor:
    """(4,4) tile values -> packed (1,16) uint8 device tensor."""
    arr = np.ascontiguousarray(np.asarray(board), dtype=np.int64)
    require_4x4(arr, 'board')
    return ops.pack(arr.reshape(1, 16))


def to_numpy(packed: torch.Tensor, dtype) -> np.ndarray:
    """packed (1,16) -> (4,4) tile values of `dtype`."""
    return ops.unpack(packed).cpu().numpy().reshape(4, 4).astype(dtype, copy=False)


def host(t: torch.Tensor):
    return t.cpu().numpy()


def rng_for(seed):
    """(seed, step_index) of the Philox stream a call with `seed` draws from."""
    if seed is None:
        return runtime.entropy_stream()
    return runtime.seed_to_u64(seed), 0
