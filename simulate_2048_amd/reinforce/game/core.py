"""
Core game logic with the interface of ``reinforce/game/core.py`` (reference lines cited per
function).  Boards are ``(4, 4)`` int32 tile values (``torch`` tensors on the runtime device, or
anything convertible); results are device tensors.  Semantics follow the JAX engine where it
differs from the NumPy one (``is_done`` of the all-zero board is True: core.py:320-321).
"""

from __future__ import annotations

import torch

from ... import ops, runtime
from ...interop import to_tensor, wrap

BOARD_SIZE = 4
NUM_ACTIONS = 4


def _board_tensor(board) -> torch.Tensor:
    rt = runtime.current()
    t = to_tensor(board)
    if t.shape[-2:] != (4, 4):
        raise ValueError(f'expected a (4, 4) board, got shape {tuple(t.shape)}')
    if t.dtype != torch.int32 or t.device != rt.device:
        t = t.to(device=rt.device, dtype=torch.int32)
    return t if t.is_contiguous() else t.contiguous()


def _packed(board) -> torch.Tensor:
    return ops.pack(_board_tensor(board).reshape(-1, 16))


def _values(packed: torch.Tensor) -> torch.Tensor:
    return wrap(ops.unpack(packed, dtype=torch.int32).reshape(4, 4))


def merge_row(row):
    """[ref core.py:43-101] merge one row to the left -> (row (4,), score int32)."""
    rt = runtime.current()
    r = to_tensor(row).to(device=rt.device, dtype=torch.int32).reshape(4)
    board = torch.zeros((4, 4), dtype=torch.int32, device=rt.device)
    board[0] = r
    after, reward, _ = ops.afterstate(_packed(board), [0])
    return _values(after)[0], wrap(reward[0])


def slide_and_merge(board):
    """[ref core.py:104-126] slide the whole board left -> (board, score int32)."""
    after, reward, _ = ops.afterstate(_packed(board), [0])
    return _values(after), wrap(reward[0])


def latent_state(state, action):
    """[ref core.py:129-170] afterstate of `action` (0 left, 1 up, 2 right, 3 down) -> (board, reward int32)."""
    after, reward, _ = ops.afterstate(_packed(state), [int(action) & 3])
    return _values(after), wrap(reward[0])


def fill_cells(state, key):
    """[ref core.py:173-231] one new tile (2 w.p. 0.9, 4 w.p. 0.1) in a uniformly chosen empty cell."""
    packed = _packed(state)
    ops.spawn_random(packed, runtime.seed_to_u64(key), 0, 1)
    return _values(packed)


def next_state(state, action, key):
    """[ref core.py:234-272] move + spawn iff the board changed -> (board, reward float32; 0.0 if unchanged)."""
    packed = _packed(state)
    out = ops.step(packed, [int(action) & 3], runtime.seed_to_u64(key), 0)
    return _values(packed), wrap(out['reward'][0].to(torch.float32))


def legal_actions_mask(state):
    """[ref core.py:275-302] bool (4,): action a changes the board. Order [left, up, right, down]."""
    bits = ops.status(_packed(state), done=False, legal=True)['legal'][0].to(torch.int32)
    return wrap(((bits >> torch.arange(4, device=bits.device, dtype=torch.int32)) & 1).to(torch.bool))


def is_done(state):
    """[ref core.py:305-321] no legal action (so the all-zero board counts as done, unlike the NumPy engine)."""
    return ~legal_actions_mask(state).any()


def create_initial_board(key):
    """[ref core.py:324-343] zeros + two sequential single spawns."""
    return _values(ops.reset(1, runtime.seed_to_u64(key)))


def encode_observation(board):
    """[ref core.py:346-377] log2(tile)/16 per cell (0 for empty), flattened to (16,) float32."""
    return wrap(ops.status(_packed(board), done=False, legal=False, obs=True)['obs'][0])


def max_tile(board):
    """[ref core.py:427-442] largest tile VALUE on the board."""
    e = ops.status(_packed(board), done=False, legal=False, max_exp=True)['max_exp'][0].to(torch.int32)
    return wrap(torch.where(e > 0, torch.ones_like(e) << e, torch.zeros_like(e)))


def count_empty(board):
    """[ref core.py:445-460] number of empty cells."""
    return wrap(ops.status(_packed(board), done=False, legal=False, n_empty=True)['n_empty'][0].to(torch.int32))
