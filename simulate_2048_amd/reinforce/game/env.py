"""
Functional environment API with the interface of ``reinforce/game/env.py`` (reference lines cited
per function): ``GameState``, ``reset``/``step``, the ``batched_*`` forms and the OOP
``Environment`` wrapper.  State fields are ``torch`` tensors on the GPU.

The batched forms are what ``reinforce/training/self_play.py:300,316-317,354`` calls once per
move; ``batched_step`` / ``batched_reset`` are ONE kernel launch (int32 value boards in, int32 value boards out) that also
writes the observation and the legal mask of the new state, so the two getters the loop calls next cost no launch.  Everything
returned is an ``interop.DeviceArray`` -- a GPU-resident torch tensor that ``np.array(...)`` copies to
the host in one go -- and every argument may be a torch tensor, a NumPy array or any ``__dlpack__`` /
``__array__`` producer (JAX ``(N, 2)`` uint32 keys, ``(N,)`` int32 actions), so the caller's loop runs
unchanged.  Randomness follows the reference's contract ``batched_step == vmap(step)``: environment i
draws from ``keys[i]`` only, so row i equals ``step(states[i], actions[i], keys[i])``.
For throughput work use ``simulate_2048_amd.BatchedEnv2048``, which keeps boards packed.
"""

from __future__ import annotations

import threading
from typing import NamedTuple

import numpy as np
import torch

from ... import capi, runtime
from ...interop import key_words, keys_on_device, to_tensor, wrap
from ...resultpool import RESET_FIELDS, STAGE_MAX_BYTES, pool, stager
from .core import _board_tensor


class GameState(NamedTuple):
    """[ref env.py:33-52] board (4,4) int32, step_count int32, done bool, total_reward float32
    (each with a leading N in the batched forms)."""

    board: torch.Tensor
    step_count: torch.Tensor
    done: torch.Tensor
    total_reward: torch.Tensor


def _host_small(x):
    """`x` as a NumPy array if it lives on the host and is small enough for the pinned staging ring, else None."""
    if isinstance(x, torch.Tensor):
        return None
    if isinstance(x, np.ndarray):
        a = x
    elif isinstance(x, (list, tuple, int, np.integer)):
        a = np.asarray(x)
    elif hasattr(x, '__dlpack__') and getattr(x, 'device', None) is not None and 'cpu' not in str(x.device).lower():
        return None  # a device array of another framework: DLPack hand-over, no host round trip
    else:
        a = np.asarray(x)
    return a if a.nbytes <= STAGE_MAX_BYTES else None


def _keys_ptr(keys, rt):
    """(device-visible pointer, n, keep-alive) of (N, 2) PRNG keys: small host arrays go through the pinned ring."""
    a = _host_small(keys)
    if a is not None and a.dtype.kind in 'iu':
        if a.ndim == 1 and a.size == 2:
            a = a.reshape(1, 2)
        if a.ndim != 2 or a.shape[1] != 2:
            raise ValueError(f'expected PRNG keys of shape (N, 2), got {a.shape}')
        if a.dtype != np.uint32:
            a = (a.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)  # keep the low 32 bits as a bit pattern
        return stager(rt.device).stage(np.ascontiguousarray(a)), a.shape[0], None
    k = keys_on_device(keys, rt.device)
    return k.data_ptr(), int(k.shape[0]), k


def _actions_ptr(actions, n, rt):
    a = _host_small(actions)
    if a is not None and a.dtype.kind in 'iub':
        if a.size != n:
            raise ValueError(f'batched_step: {n} environments but {a.size} actions')
        return stager(rt.device).stage(np.ascontiguousarray(a.reshape(n), dtype=np.int32)), None
    t = _dev(actions, torch.int32, n, rt)
    return t.data_ptr(), t


class _InlineScratch(threading.local):
    """Per-thread host arrays whose addresses are known: where the keys / actions of a small move are put (one NumPy
    assignment each, which also converts dtype and layout) for env2048_step_state_i32_keyed_inline to read during the call."""

    def __init__(self):
        self.keys = np.zeros((capi.INLINE_MAX_ENVS, 2), dtype=np.uint32)
        self.keys_i32 = self.keys.view(np.int32)
        self.actions = np.zeros((capi.INLINE_MAX_ENVS,), dtype=np.int32)
        self.keys_addr = self.keys.ctypes.data
        self.actions_addr = self.actions.ctypes.data


_inline = _InlineScratch()


def _inline_inputs(keys, actions, n):
    """(host address of n x 2 uint32 keys, host address of n int32 actions) when both are small HOST arrays -- the shape of the
    reference's move loop (self_play.py:333-354: keys split and actions picked on the host for N = 2 ... 8 games) -- else None.
    Raises the same ValueErrors as the staged path for wrong shapes."""
    if n > capi.INLINE_MAX_ENVS:
        return None
    k = _host_small(keys)
    if k is None or k.dtype.kind not in 'iu':
        return None
    a = _host_small(actions)
    if a is None or a.dtype.kind not in 'iub':
        return None
    if k.ndim == 1 and k.size == 2:
        k = k.reshape(1, 2)
    if k.ndim != 2 or k.shape[1] != 2:
        raise ValueError(f'expected PRNG keys of shape (N, 2), got {k.shape}')
    if k.shape[0] != n:
        raise ValueError(f'batched_step: {n} environments but {k.shape[0]} keys')
    if a.size != n:
        raise ValueError(f'batched_step: {n} environments but {a.size} actions')
    s = _inline
    if k.dtype == np.uint32:
        s.keys[:n] = k
    elif k.dtype == np.int32:
        s.keys_i32[:n] = k  # the same 32 bits
    else:
        s.keys[:n] = k.astype(np.int64) & 0xFFFFFFFF  # keep the low 32 bits as a bit pattern
    s.actions[:n] = a.reshape(n)
    return s.keys_addr, s.actions_addr


def _state_ptrs(states: GameState, rt):
    """Pointers of a state's four fields + n + keep-alive list.  Tensors that ARE the tensors of a pooled result set (the
    state came out of batched_step / batched_reset) are recognised by identity: nothing to check, nothing to convert."""
    hit = pool.lookup(states.board)
    if hit is not None and hit[1] == 'board' and not hit[3]:
        rs = hit[0]
        if states.step_count is rs.step_count and states.done is rs.done and states.total_reward is rs.total:
            p = rs.ptr
            return p['board'], p['step_count'], p['done'], p['total'], rs.n, None
    board_in = _board_tensor(states.board).reshape(-1, 4, 4)
    n = board_in.shape[0]
    steps_in = _dev(states.step_count, torch.int32, n, rt)
    total_in = _dev(states.total_reward, torch.float32, n, rt)
    d = to_tensor(states.done)
    if d.dtype == torch.bool and d.device == rt.device:
        done_in = d.reshape(n).contiguous().view(torch.uint8)  # same bytes, no copy
    else:
        done_in = _dev(d.to(torch.uint8), torch.uint8, n, rt)
    return board_in.data_ptr(), steps_in.data_ptr(), done_in.data_ptr(), total_in.data_ptr(), n, (board_in, steps_in, total_in, done_in)


def batched_reset(keys) -> GameState:
    """[ref env.py:176-190] fresh games, one per key row: keys (N, 2) -> GameState with leading N.
    One launch (env2048_reset_state_i32_keyed); board i depends on keys[i] only (= reset(keys[i])).  The observation and
    the legal mask of the fresh boards are written by the same launch and kept with the state (see batched_get_observation)."""
    rt = runtime.current()
    kptr, n, keep = _keys_ptr(keys, rt)
    dev, stream = rt.handles()
    rs = pool.acquire(n, rt.device, stream or 0)
    try:
        p = rs.ptr
        if rs.mirror_delta:  # pooled set with a pinned host mirror: the launch writes it and (small batches) signals completion
            rt.lib.reset_state_i32_keyed_signalled(dev, stream, n, kptr, p['board'], p['step_count'], p['done'], p['total'], p['obs'],
                                                   p['legal'], rs.mirror_delta, rs.seq_ptr, rs.next_serial())
            rs.mark_written(True, rs.seq_ptr is not None, RESET_FIELDS)
        else:
            rt.lib.reset_state_i32_keyed(dev, stream, n, kptr, p['board'], p['step_count'], p['done'], p['total'], p['obs'], p['legal'], 0)
            rs.mark_written(fields=RESET_FIELDS)
        return GameState(board=rs.board, step_count=rs.step_count, done=rs.done, total_reward=rs.total)
    finally:
        rs.leased = False  # (the state being returned holds the tensors by now; after an error nobody does and the set is free)


def _dev(x, dtype, n, rt) -> torch.Tensor:
    """`x` as a contiguous (n,) device tensor of `dtype`; no copy when it already is one."""
    t = to_tensor(x)
    if t.dtype != dtype or t.device != rt.device:
        t = t.to(device=rt.device, dtype=dtype)
    if t.dim() == 1 and t.shape[0] == n and t.is_contiguous():
        return t
    return t.reshape(n).contiguous()


def batched_step(states: GameState, actions, keys):
    """[ref env.py:193-211] one move per environment -> (GameState, reward (N,) f32, done (N,) bool, info).

    ONE fused, out-of-place kernel (env2048_step_state_i32_keyed_signalled): the input state is only read (functional API), the
    new state, reward, done, max_tile AND the observation / legal mask of the new state are written by the same launch into a
    result set that is reused once the caller has let go of it (``resultpool``) -- a move of the reference's self-play loop
    (self_play.py:316-317,354: observation, legal mask, step) is one launch and creates no tensor.  Finished games are frozen
    exactly like the reference's done-guard (env.py:103-104): state unchanged, reward 0.0, done True, step_count not advanced.
    Environment i draws its tile from keys[i] alone (vmap semantics).
    """
    rt = runtime.current()
    b_in, s_in, d_in, t_in, n, keep_state = _state_ptrs(states, rt)
    dev, stream = rt.handles()
    rs = pool.acquire(n, rt.device, stream or 0)  # (never the set the input state lives in: the caller still holds that one)
    try:
        p = rs.ptr
        inline = _inline_inputs(keys, actions, n) if rs.mirror_delta else None
        if inline is not None:  # host-side keys / actions of a small move ride in the kernel-argument block: nothing to stage
            rt.lib.step_state_i32_keyed_inline(dev, stream, n, inline[0], b_in, p['board'], inline[1], s_in, p['step_count'], d_in, p['done'],
                                               t_in, p['total'], p['reward'], p['max_tile'], p['obs'], p['legal'], rs.mirror_delta, rs.seq_ptr,
                                               rs.next_serial())
            rs.mark_written(True, rs.seq_ptr is not None)
            new_state = GameState(board=rs.board, step_count=rs.step_count, done=rs.done, total_reward=rs.total)
            return new_state, rs.reward, rs.done, {'max_tile': rs.max_tile, 'step_count': rs.step_count, 'total_reward': rs.total}
        kptr, nk, keep_keys = _keys_ptr(keys, rt)
        if nk != n:
            raise ValueError(f'batched_step: {n} environments but {nk} keys')
        aptr, keep_actions = _actions_ptr(actions, n, rt)
        if rs.mirror_delta:
            rt.lib.step_state_i32_keyed_signalled(dev, stream, n, kptr, b_in, p['board'], aptr, s_in, p['step_count'], d_in, p['done'], t_in,
                                                  p['total'], p['reward'], p['max_tile'], p['obs'], p['legal'], rs.mirror_delta, rs.seq_ptr,
                                                  rs.next_serial())
            rs.mark_written(True, rs.seq_ptr is not None)
        else:
            rt.lib.step_state_i32_keyed_mirrored(dev, stream, n, kptr, b_in, p['board'], aptr, s_in, p['step_count'], d_in, p['done'], t_in,
                                                 p['total'], p['reward'], p['max_tile'], p['obs'], p['legal'], 0)
            rs.mark_written()
        new_state = GameState(board=rs.board, step_count=rs.step_count, done=rs.done, total_reward=rs.total)
        info = {'max_tile': rs.max_tile, 'step_count': rs.step_count, 'total_reward': rs.total}
        return new_state, rs.reward, rs.done, info
    finally:
        rs.leased = False


def _stashed(states: GameState, field: str):
    """obs / legal written by the launch that produced `states`, if the state is a pooled one and still as written."""
    hit = pool.lookup(states.board)
    if hit is not None and hit[1] == 'board' and not hit[3] and hit[0].stash_valid():
        return getattr(hit[0], field)
    return None


def batched_get_observation(states: GameState) -> torch.Tensor:
    """[ref env.py:214-228] (N, 16) float32 log2/16 observations.  For a state that came out of batched_step / batched_reset
    this is the array the same launch already wrote (no kernel); any other board gets one status kernel."""
    obs = _stashed(states, 'obs')
    if obs is not None:
        return obs
    rt = runtime.current()
    boards = _board_tensor(states.board).reshape(-1, 16)
    n = boards.shape[0]
    obs = torch.empty((n, 16), dtype=torch.float32, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.status_i32(dev, stream, n, boards.data_ptr(), obs.data_ptr(), None, None)
    return wrap(obs)


def batched_get_legal_actions(states: GameState) -> torch.Tensor:
    """[ref env.py:231-245] (N, 4) bool legal-action masks (stashed by the producing launch, else one status kernel)."""
    legal = _stashed(states, 'legal')
    if legal is not None:
        return legal
    rt = runtime.current()
    boards = _board_tensor(states.board).reshape(-1, 16)
    n = boards.shape[0]
    legal = torch.empty((n, 4), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.status_i32(dev, stream, n, boards.data_ptr(), None, legal.data_ptr(), None)
    return wrap(legal.view(torch.bool))


def _squeeze(state: GameState) -> GameState:
    return GameState(state.board[0], state.step_count[0], state.done[0], state.total_reward[0])


def _unsqueeze(state: GameState) -> GameState:
    rt = runtime.current()
    return GameState(
        _board_tensor(state.board).reshape(1, 4, 4),
        to_tensor(state.step_count).to(device=rt.device, dtype=torch.int32).reshape(1),
        to_tensor(state.done).to(device=rt.device, dtype=torch.bool).reshape(1),
        to_tensor(state.total_reward).to(device=rt.device, dtype=torch.float32).reshape(1),
    )


def _one_key(key) -> np.ndarray:
    """A single PRNG key ((2,) uint32 array-like or a plain integer seed) as a (1, 2) uint32 HOST array (so that it travels in
    the kernel arguments like the keys of a small batch: no staging, no copy to the device)."""
    k0, k1 = key_words(key)
    return np.array([[k0, k1]], dtype=np.uint32)


def _single_set(state: GameState):
    """The one-environment result set whose un-batched aliases make up `state` (it came out of reset / step), else None."""
    hit = pool.lookup(state.board)
    if hit is not None and hit[3]:
        single = hit[0].single
        if state.step_count is single['step_count'] and state.done is single['done'] and state.total_reward is single['total']:
            return hit[0]
    return None


def _single_state(batched: GameState) -> GameState:
    """Un-batched view of a one-environment state: the set's own alias objects when it is a pooled set (nothing is created),
    else plain indexing."""
    hit = pool.lookup(batched.board)
    if hit is not None and hit[0].single is not None:
        single = hit[0].single
        return GameState(single['board'], single['step_count'], single['done'], single['total'])
    return _squeeze(batched)


def reset(key) -> GameState:
    """[ref env.py:55-76] new game: board with two tiles, step_count 0, done False, total_reward 0."""
    return _single_state(batched_reset(_one_key(key)))


def step(state: GameState, action, key):
    """[ref env.py:79-133] -> (GameState, reward f32, done bool, info dict).  A state that came out of reset / step is
    recognised by identity: the move is the batched launch on its one-environment result set (host key and action in the
    kernel arguments, results mirrored to the host), no view or conversion in between."""
    rs = _single_set(state)
    batched = GameState(rs.board, rs.step_count, rs.done, rs.total) if rs is not None else _unsqueeze(state)
    new_state, reward, done, info = batched_step(batched, np.array([int(action)], dtype=np.int32), _one_key(key))
    hit = pool.lookup(new_state.board)
    if hit is not None and hit[0].single is not None:
        single = hit[0].single
        return (GameState(single['board'], single['step_count'], single['done'], single['total']), single['reward'], single['done'],
                {'max_tile': single['max_tile'], 'step_count': single['step_count'], 'total_reward': single['total']})
    return _squeeze(new_state), reward[0], done[0], {k: v[0] for k, v in info.items()}


def get_observation(state: GameState) -> torch.Tensor:
    """[ref env.py:136-151] (16,) float32."""
    rs = _single_set(state)
    if rs is not None and rs.stash_valid():
        return rs.single['obs']  # written by the launch that produced the state
    return batched_get_observation(_unsqueeze(state))[0]


def get_legal_actions(state: GameState) -> torch.Tensor:
    """[ref env.py:154-169] (4,) bool."""
    rs = _single_set(state)
    if rs is not None and rs.stash_valid():
        return rs.single['legal']
    return batched_get_legal_actions(_unsqueeze(state))[0]


class Environment:
    """[ref env.py:248-304] stateful convenience wrapper; splits its own key per call."""

    def __init__(self, seed: int = 0):
        self.key = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._draws = 0
        self.state = self.reset_env()

    def _subkey(self):
        # ##>: stand-in for jax.random.split: a fresh 64-bit key per call, derived from (seed, draw index)
        self._draws += 1
        mixed = (self.key * 0x9E3779B97F4A7C15 + self._draws * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        return [mixed >> 32, mixed & 0xFFFFFFFF]

    def reset_env(self) -> GameState:
        self.state = reset(self._subkey())
        return self.state

    def step_env(self, action: int):
        self.state, reward, done, info = step(self.state, action, self._subkey())
        return self.state, float(reward), bool(done), info

    def observation(self) -> torch.Tensor:
        return get_observation(self.state)

    def legal_actions(self) -> torch.Tensor:
        return get_legal_actions(self.state)

    @property
    def board(self) -> torch.Tensor:
        return self.state.board

    @property
    def done(self) -> bool:
        return bool(self.state.done)
