"""
REFERENCE-SHAPED CPU BASELINE -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT.

A per-board NumPy restatement of the transition with the SAME algorithmic structure as the reference's NumPy engine
(one board at a time as a (4, 4) int64 array of tile VALUES; a quarter-turn per action, a Python loop over the four lines,
boolean-mask compaction, ``argwhere`` for the empty cells): what "the reference's NumPy step on this host" costs, for the
``cpu_baseline`` leg of ``bench.py`` on the GPU box, where the reference itself cannot be present.  It is written from the
behaviour documented in SURVEY.md App. A (not from the reference's text) and checked against the reference-generated
fixtures in ``tests/test_oracle_golden.py``.  The spawn choice (k-th empty cell in row-major order, tile value) is
injected from the build's Philox step word, exactly as ``oracle/gen_golden.py`` injects it into the reference.

Reference lines this follows: twentyfortyeight/core/gameboard.py:23-69 (line rule), :72-102 (board = four lines),
:105-129 (rotate / slide / rotate back), :291-325 (transition: validity test, spawn iff changed, reward), :328-348
(terminal test); twentyfortyeight/core/gamemove.py:132-164 (validity of a left move).
"""

from __future__ import annotations

import numpy as np


def merge_line(line: np.ndarray) -> tuple[int, np.ndarray]:
    """One line toward its start: zeros dropped, equal neighbours merged once from the start outward [ref :23-69]."""
    tiles = line[line != 0]
    if tiles.size <= 1:
        return 0, tiles
    out, gained, pos = [], 0, 0
    last = tiles.size - 1
    while pos < last:
        if tiles[pos] == tiles[pos + 1]:
            out.append(tiles[pos] * 2)
            gained += int(tiles[pos]) * 2
            pos += 2
        else:
            out.append(tiles[pos])
            pos += 1
    if pos == last:
        out.append(tiles[last])
    return gained, np.asarray(out, dtype=line.dtype)


def slide_left(board: np.ndarray) -> tuple[int, np.ndarray]:
    """[ref :72-102] every row through merge_line, padded with zeros on the far side."""
    moved = np.zeros_like(board)
    gained = 0
    for r in range(board.shape[0]):
        g, merged = merge_line(board[r])
        gained += g
        moved[r, : merged.size] = merged
    return gained, moved


def left_move_possible(board: np.ndarray) -> bool:
    """[ref gamemove.py:132-164] an empty cell before a tile, or two equal neighbours, in some row."""
    a, b = board[:, :-1], board[:, 1:]
    return bool(((a == 0) & (b != 0)).any() or ((a != 0) & (a == b)).any())


def afterstate(board: np.ndarray, action: int) -> tuple[np.ndarray, int]:
    """[ref :105-129] quarter-turns so that the move becomes 'left', slide, turn back."""
    turned = np.rot90(board, k=action)
    gained, moved = slide_left(turned)
    return np.rot90(moved, k=-action), gained


def terminal(board: np.ndarray) -> bool:
    """[ref :328-348] full, and no equal vertical or horizontal neighbours."""
    return bool((board != 0).all() and not (board[:-1] == board[1:]).any() and not (board[:, :-1] == board[:, 1:]).any())


def transition(board: np.ndarray, action: int, word: int) -> tuple[np.ndarray, int, bool]:
    """[ref :291-325 + :328-348] one environment step with the spawn taken from the build's step word `word`
    (include/env2048.h: m = word & ~3, P = m * n_empty, cell rank P >> 32 in row-major order, tile 4 iff
    P mod 2^32 < floor(0.1 * 2^32)).  Returns (board', reward, done); an invalid move returns the board itself."""
    action &= 3
    turned = np.rot90(board, k=action)
    if not left_move_possible(turned):
        return board, 0, terminal(board)
    gained, moved = slide_left(turned)
    new = np.rot90(moved, k=-action).copy()
    empties = np.argwhere(new == 0)
    product = (word & 0xFFFFFFFC) * len(empties)
    cell = empties[product >> 32]
    new[cell[0], cell[1]] = 4 if (product & 0xFFFFFFFF) < 429496729 else 2
    return new, gained, terminal(new)
