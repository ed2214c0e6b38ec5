"""Shared helpers of the NumPy-facing mirrors."""

from __future__ import annotations

import numpy as np

from ... import runtime


MAX_SIZE = 8  # boards up to 8 x 8 (env2048_nxn_*, ENV2048_NXN_MAX_SIZE); 4 x 4 takes the packed kernels


def require_4x4(board: np.ndarray, what: str) -> None:
    if board.ndim != 2 or board.shape != (4, 4):
        raise ValueError(
            f'{what}: the MI355X engine handles 4x4 boards only (SURVEY.md App. A.5), got shape {board.shape}'
        )


def is_4x4(board: np.ndarray, what: str) -> bool:
    """True for the packed 4 x 4 path, False for the general S x S path (2 <= S <= 8); anything else is refused."""
    if board.ndim == 2 and board.shape == (4, 4):
        return True
    if board.ndim != 2 or board.shape[0] != board.shape[1] or not 2 <= board.shape[0] <= MAX_SIZE:
        raise ValueError(f'{what}: the MI355X engine handles square boards of size 2 ... {MAX_SIZE}, got shape {board.shape}')
    return False


def rng_for(seed, count: int = 1):
    """(seed, first step_index) of the Philox stream a call with `seed` draws from; ``count`` = how many
    consecutive step words the call consumes (reserved on the process-wide entropy stream when unseeded)."""
    if seed is None:
        return runtime.entropy_stream(count)
    return runtime.seed_to_u64(seed), 0
