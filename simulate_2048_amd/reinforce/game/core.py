"""
Core game logic with the interface of ``reinforce/game/core.py`` (reference lines cited per
function).  Boards are ``(4, 4)`` int32 tile values (``torch`` tensors on the runtime device, or
anything convertible); results are device tensors.  Semantics follow the JAX engine where it
differs from the NumPy one (``is_done`` of the all-zero board is True: core.py:320-321).
"""

from __future__ import annotations

import threading

import numpy as np
import torch

from ... import ops, runtime
from ...interop import key_words, to_tensor, wrap
from ...resultpool import NEXT_STATE_FIELDS, pool

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


# ---- one launch per move for the reference's single-game loops --------------------------------------------------------------
# play_game / evaluate_games (reinforce/training/self_play.py:95-135, reinforce/evaluate.py:96-128) call, per move,
# encode_observation(board), legal_actions_mask(board), next_state(board, action, key), is_done(new_board) and max_tile(board).
# next_state runs the fused move kernel (env2048_step_state_i32_keyed_inline: key and action travel in the kernel arguments)
# into a pooled one-environment result set, which also receives the observation, legal mask, terminal flag and max tile of the
# NEW board; the board it returns is that set's (4, 4) alias, so the four queries about it are answered from the set --
# recognised by identity, valid while torch's version counter says nobody edited it in place -- without a launch, and their
# np.array / bool / float read-backs come from the set's host mirror.  Any other board takes one status launch per query.
class _MoveScratch(threading.local):
    def __init__(self):
        self.key = np.zeros((1, 2), dtype=np.uint32)
        self.action = np.zeros((1,), dtype=np.int32)
        self.key_addr, self.action_addr = self.key.ctypes.data, self.action.ctypes.data
        self.not_done = {}  # device -> a zero byte there (the "not finished" flag next_state's kernel reads)


_move = _MoveScratch()


def _alias_set(board, field: str):
    """The one-environment result set `board` is the (4, 4) alias of, if `field` of that set is current; else None."""
    if isinstance(board, torch.Tensor):
        hit = pool.lookup(board)
        if hit is not None and hit[3] and hit[1] == 'board' and hit[0].has(field):
            return hit[0]
    return None


def _status_i32(board, *, obs=False, legal=False, done=False):
    """One env2048_status_i32 launch on a (4, 4) value board that is not a pooled alias -> the requested tensors."""
    rt = runtime.current()
    b = _board_tensor(board)
    dev, stream = rt.handles()
    o = torch.empty((16,), dtype=torch.float32, device=rt.device) if obs else None
    l4 = torch.empty((4,), dtype=torch.uint8, device=rt.device) if legal else None
    d = torch.empty((1,), dtype=torch.uint8, device=rt.device) if done else None
    rt.lib.status_i32(dev, stream, 1, b.data_ptr(), None if o is None else o.data_ptr(), None if l4 is None else l4.data_ptr(),
                      None if d is None else d.data_ptr())
    return o, l4, d


def _afterstate_i32(board: torch.Tensor, action: int):
    """One launch of env2048_afterstate_i32 on one int32 value board (the move rides in the kernel arguments) -> (board, score)."""
    rt = runtime.current()
    dev, stream = rt.handles()
    if board.data_ptr() & 15:  # (a view at an odd offset of somebody's tensor)
        board = board.clone()
    out = torch.empty((20,), dtype=torch.int32, device=board.device)  # 16 cells + the score (16-byte aligned behind them)
    rt.lib.afterstate_i32(dev, stream, 1, board.data_ptr(), None, action & 3, out.data_ptr(), out.data_ptr() + 64, None)
    return out[:16].view(4, 4), out[16]


def merge_row(row):
    """[ref core.py:43-101] merge one row to the left -> (row (4,), score int32)."""
    rt = runtime.current()
    r = to_tensor(row).to(device=rt.device, dtype=torch.int32).reshape(4)
    after, reward = _afterstate_i32(torch.nn.functional.pad(r, (0, 12)), 0)  # the row in an otherwise empty board
    return wrap(after[0]), wrap(reward)


def slide_and_merge(board):
    """[ref core.py:104-126] slide the whole board left -> (board, score int32)."""
    after, reward = _afterstate_i32(_board_tensor(board), 0)
    return wrap(after), wrap(reward)


def latent_state(state, action):
    """[ref core.py:129-170] afterstate of `action` (0 left, 1 up, 2 right, 3 down) -> (board, reward int32)."""
    after, reward = _afterstate_i32(_board_tensor(state), int(action))
    return wrap(after), wrap(reward)


def fill_cells(state, key):
    """[ref core.py:173-231] one new tile (2 w.p. 0.9, 4 w.p. 0.1) in a uniformly chosen empty cell."""
    packed = _packed(state)
    ops.spawn_random(packed, runtime.seed_to_u64(key), 0, 1)
    return _values(packed)


def next_state(state, action, key):
    """[ref core.py:234-272] move + spawn iff the board changed -> (board, reward float32; 0.0 if unchanged).
    ONE launch, which also leaves the observation, legal mask, terminal flag and max tile of the new board with it (above)."""
    rt = runtime.current()
    src = _alias_set(state, 'board')
    keep = None
    if src is not None:
        board_ptr = src.ptr['board']
    else:
        keep = _board_tensor(state)
        board_ptr = keep.data_ptr()
    dev, stream = rt.handles()
    not_done = _move.not_done.get(rt.device)
    if not_done is None:
        not_done = _move.not_done[rt.device] = torch.zeros((16,), dtype=torch.uint8, device=rt.device)
    rs = pool.acquire(1, rt.device, stream or 0)  # (never the set `state` lives in: the caller still holds that one)
    try:
        k0, k1 = key_words(key)
        _move.key[0, 0], _move.key[0, 1] = k0, k1
        _move.action[0] = int(action) & 3
        p = rs.ptr
        rt.lib.step_state_i32_keyed_inline(dev, stream, 1, _move.key_addr, board_ptr, p['board'], _move.action_addr, None, None,
                                           not_done.data_ptr(), p['done'], None, None, p['reward'], p['max_tile'], p['obs'], p['legal'],
                                           rs.mirror_delta, rs.seq_ptr if rs.mirror_delta else None, rs.next_serial())
        rs.mark_written(bool(rs.mirror_delta), bool(rs.mirror_delta) and rs.seq_ptr is not None, NEXT_STATE_FIELDS)
        if rs.single is not None:
            return rs.single['board'], rs.single['reward']
        return wrap(rs.board[0]), wrap(rs.reward[0])
    finally:
        rs.leased = False


def legal_actions_mask(state):
    """[ref core.py:275-302] bool (4,): action a changes the board. Order [left, up, right, down]."""
    rs = _alias_set(state, 'legal')
    if rs is not None:
        return rs.single['legal']
    return wrap(_status_i32(state, legal=True)[1].view(torch.bool))


def is_done(state):
    """[ref core.py:305-321] no legal action (so the all-zero board counts as done, unlike the NumPy engine)."""
    rs = _alias_set(state, 'done')
    if rs is not None:
        return rs.single['done']
    return wrap(_status_i32(state, done=True)[2].view(torch.bool)[0])


def create_initial_board(key):
    """[ref core.py:324-343] zeros + two sequential single spawns."""
    return _values(ops.reset(1, runtime.seed_to_u64(key)))


def encode_observation(board):
    """[ref core.py:346-377] log2(tile)/16 per cell (0 for empty), flattened to (16,) float32."""
    rs = _alias_set(board, 'obs')
    if rs is not None:
        return rs.single['obs']
    return wrap(_status_i32(board, obs=True)[0])


def max_tile(board):
    """[ref core.py:427-442] largest tile VALUE on the board."""
    rs = _alias_set(board, 'max_tile')
    if rs is not None:
        return rs.single['max_tile']
    return wrap(_board_tensor(board).max())  # (a board nobody's launch has described: one reduction over its 16 values)


def count_empty(board):
    """[ref core.py:445-460] number of empty cells."""
    return wrap(torch.sum(_board_tensor(board) == 0, dtype=torch.int32))
