"""Shared helpers of the NumPy-facing mirrors."""

from __future__ import annotations

import numpy as np

from ... import runtime


def require_4x4(board: np.ndarray, what: str) -> None:
    if board.ndim != 2 or board.shape != (4, 4):
        raise ValueError(
            f'{what}: the MI355X engine handles 4x4 boards only (SURVEY.md App. A.5), got shape {board.shape}'
        )


def rng_for(seed, count: int = 1):
    """(seed, first step_index) of the Philox stream a call with `seed` draws from; ``count`` = how many
    consecutive step words the call consumes (reserved on the process-wide entropy stream when unseeded)."""
    if seed is None:
        return runtime.entropy_stream(count)
    return runtime.seed_to_u64(seed), 0
