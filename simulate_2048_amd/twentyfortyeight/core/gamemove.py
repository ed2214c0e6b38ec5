"""
Game move utilities: legal / illegal moves.  Mirror of ``twentyfortyeight/core/gamemove.py``
(reference lines cited per function); the adjacency tests run in the status kernel.
"""

from __future__ import annotations

from numpy import asarray, ndarray, rot90

from ... import scalar
from ._device import is_4x4


def _legal_bits(state: ndarray) -> int:
    state = asarray(state)
    ops = scalar.current()
    return (ops.status if is_4x4(state, 'board') else ops.status_nxn)(state)[1]


def _can_move_direction(board: ndarray, direction: int) -> bool:
    """[ref gamemove.py:9-42] move possible in `direction` (0 left, 1 up, 2 right, anything else down)."""
    bit = direction if direction in (0, 1, 2) else 3
    return bool((_legal_bits(board) >> bit) & 1)


def legal_actions_mask(state: ndarray) -> tuple[bool, bool, bool, bool]:
    """[ref gamemove.py:45-83] mask for (left, up, right, down), True = legal."""
    bits = _legal_bits(state)
    return (bool(bits & 1), bool(bits & 2), bool(bits & 4), bool(bits & 8))


def illegal_actions(state: ndarray) -> list[int]:
    """[ref gamemove.py:86-106]"""
    mask = legal_actions_mask(state)
    return [i for i in range(4) if not mask[i]]


def legal_actions(state: ndarray) -> list[int]:
    """[ref gamemove.py:109-129]"""
    mask = legal_actions_mask(state)
    return [i for i in range(4) if mask[i]]


def can_move(board: ndarray) -> bool:
    """[ref gamemove.py:132-164] can any tile move LEFT on this (possibly rotated) board."""
    return bool(_legal_bits(board) & 1)


__all__ = ['legal_actions_mask', 'illegal_actions', 'legal_actions', 'can_move', 'rot90']
