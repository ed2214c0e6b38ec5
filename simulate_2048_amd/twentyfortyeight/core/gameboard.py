"""
Core 2048 board functions.  Mirror of ``twentyfortyeight/core/gameboard.py``: same names, argument
meaning, return shapes/dtypes and error behaviour; the arithmetic runs in the HIP kernels
(afterstate / spawn / status / step) through the C-ABI.  Reference lines are cited per function.

Boards: 4 x 4 takes the packed kernels; any other square size from 2 to 8 takes the general kernels (``env2048_nxn_*``: the
reference's functions take any shape, its env has a ``size`` argument, its timing script runs 4 / 6 / 8).  The one difference that
cannot be avoided: the random stream is Philox (``seed`` selects a reproducible stream, ``None`` an
OS-entropy one) instead of NumPy PCG64.
"""

from __future__ import annotations

import numpy as np
from numpy import argwhere, ndarray

from ... import scalar
from ._device import MAX_SIZE, is_4x4, rng_for

# ##>: Tile spawn probabilities (90% for 2, 10% for 4) [ref gameboard.py:13].
TILE_SPAWN_PROBS: dict[int, float] = {2: 0.9, 4: 0.1}


def merge_column(column: ndarray) -> tuple[int, ndarray]:
    """[ref gameboard.py:23-69] drop zeros, merge equal neighbours once, left to right.

    Returns (score, merged column WITHOUT padding) like the reference.
    """
    column = np.asarray(column)
    if column.ndim != 1 or not 2 <= column.shape[0] <= MAX_SIZE:
        raise ValueError(f'merge_column: the MI355X engine handles columns of length 2 ... {MAX_SIZE}, got shape {column.shape}')
    size = column.shape[0]
    board = np.zeros((size, size), dtype=np.int64)
    board[0] = column
    ops = scalar.current()
    after, reward, _ = ops.afterstate(board, 0, column.dtype) if size == 4 else ops.afterstate_nxn(board, 0, column.dtype)
    row = after[0]
    return reward, row[row != 0]


def slide_and_merge(board: ndarray) -> tuple[float, ndarray]:
    """[ref gameboard.py:72-102] slide every row left and merge; score is a float like the reference."""
    board = np.asarray(board)
    ops = scalar.current()
    after, reward, _ = ops.afterstate(board, 0, board.dtype) if is_4x4(board, 'board') else ops.afterstate_nxn(board, 0, board.dtype)
    return 0.0 + float(reward), after


def latent_state(state: ndarray, action: int) -> tuple[ndarray, float]:
    """[ref gameboard.py:105-129] afterstate of `action` (0 left, 1 up, 2 right, 3 down), no new tile."""
    state = np.asarray(state)
    ops = scalar.current()
    after, reward, _ = (ops.afterstate if is_4x4(state, 'board') else ops.afterstate_nxn)(state, int(action), state.dtype)
    return after, 0.0 + float(reward)


def after_state(state: ndarray) -> list[tuple[ndarray, float]]:
    """[ref gameboard.py:132-171] every (cell, value) spawn outcome with its probability, enumerated
    on device in the reference's order (empty cells row-major, value 2 then 4)."""
    state = np.asarray(state)
    num_empty_cells = int((state == 0).sum())
    if num_empty_cells == 0:
        return [(state, 1.0)]
    ops = scalar.current()
    values, count = (ops.chance_outcomes if is_4x4(state, 'board') else ops.chance_outcomes_nxn)(state, state.dtype)
    # ##>: probabilities in float64 exactly as the reference computes them (TILE_SPAWN_PROBS[v] / n)
    return [(values[i], TILE_SPAWN_PROBS[2 if i % 2 == 0 else 4] / num_empty_cells) for i in range(count)]


def after_state_lazy(state: ndarray) -> tuple[ndarray, list[tuple[int, int]], int]:
    """[ref gameboard.py:174-202] (state reference, empty cells row-major, their count); nothing computed."""
    empty_cells = argwhere(state == 0)
    return state, [(int(c[0]), int(c[1])) for c in empty_cells], len(empty_cells)


def generate_outcome(state: ndarray, cell: tuple[int, int], value: int, num_empty: int) -> tuple[ndarray, float]:
    """[ref gameboard.py:205-244] copy of `state` with `value` at `cell`, probability P(value)/num_empty."""
    if num_empty <= 0:
        raise ValueError(f'num_empty must be > 0, got {num_empty}')
    new_state = state.copy()
    new_state[cell] = value
    return new_state, TILE_SPAWN_PROBS[value] / num_empty


def fill_cells(state: ndarray, number_tile: int, seed: int | None = None) -> ndarray:
    """[ref gameboard.py:247-288] add `number_tile` tiles (2 w.p. 0.9, 4 w.p. 0.1) to random empty
    cells.  **Mutates `state` in place** and returns it, like the reference."""
    num_available = int((state == 0).sum())
    if num_available == 0:
        return state
    number_tile = min(int(number_tile), num_available)
    ops = scalar.current()
    spawn = ops.spawn_random if is_4x4(np.asarray(state), 'board') else ops.spawn_random_nxn
    stream_seed, step = rng_for(seed, number_tile)
    state[...] = spawn(state, stream_seed, step, number_tile, state.dtype)
    return state


def next_state(state: ndarray, action: int, seed: int | None = None) -> tuple[ndarray, float]:
    """[ref gameboard.py:291-325] full transition: move, then one new tile iff the board changed.

    An invalid move returns the SAME `state` object and the int 0, like the reference (:325).
    """
    after, reward, _ = _transition(state, action, seed)
    if np.array_equal(after, state):
        return state, 0
    return after, 0.0 + float(reward)


def _transition(state: ndarray, action: int, seed: int | None, onehot_width: int = 0):
    """One launch of the step kernel on one board: (new values, reward int, done bool); with `onehot_width` also the new board's
    one-hot rows (cells, width) from a second launch behind it, both waited for once."""
    arr = np.asarray(state)
    ops = scalar.current()
    step_fn = ops.step if is_4x4(arr, 'board') else ops.step_nxn
    stream_seed, step = rng_for(seed)
    return step_fn(arr, int(action), stream_seed, step, arr.dtype, onehot_width)


def is_done(state: ndarray) -> bool:
    """[ref gameboard.py:328-348] no empty cell and no equal horizontal / vertical neighbours."""
    state = np.asarray(state)
    ops = scalar.current()
    return (ops.status if is_4x4(state, 'board') else ops.status_nxn)(state)[0]
