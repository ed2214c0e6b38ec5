"""
Functional environment API with the interface of ``reinforce/game/env.py`` (reference lines cited
per function): ``GameState``, ``reset``/``step``, the ``batched_*`` forms and the OOP
``Environment`` wrapper.  State fields are ``torch`` tensors on the GPU.

The batched forms are what ``reinforce/training/self_play.py:300,316-317,354`` calls once per
move; each is ONE kernel launch here (int32 value boards in, int32 value boards out).  Everything
returned is an ``interop.DeviceArray`` -- a GPU-resident torch tensor that ``np.array(...)`` copies to
the host in one go -- and every argument may be a torch tensor, a NumPy array or any ``__dlpack__`` /
``__array__`` producer (JAX ``(N, 2)`` uint32 keys, ``(N,)`` int32 actions), so the caller's loop runs
unchanged.  Randomness follows the reference's contract ``batched_step == vmap(step)``: environment i
draws from ``keys[i]`` only, so row i equals ``step(states[i], actions[i], keys[i])``.
For throughput work use ``simulate_2048_amd.BatchedEnv2048``, which keeps boards packed.
"""

from __future__ import annotations

from typing import NamedTuple

import torch

from ... import runtime
from ...interop import key_words, keys_on_device, to_tensor, wrap
from .core import _board_tensor


class GameState(NamedTuple):
    """[ref env.py:33-52] board (4,4) int32, step_count int32, done bool, total_reward float32
    (each with a leading N in the batched forms)."""

    board: torch.Tensor
    step_count: torch.Tensor
    done: torch.Tensor
    total_reward: torch.Tensor


def batched_reset(keys) -> GameState:
    """[ref env.py:176-190] fresh games, one per key row: keys (N, 2) -> GameState with leading N.
    One launch (env2048_reset_keyed_i32); board i depends on keys[i] only (= reset(keys[i]))."""
    rt = runtime.current()
    k = keys_on_device(keys, rt.device)
    n = int(k.shape[0])
    board = torch.empty((n, 4, 4), dtype=torch.int32, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.reset_keyed_i32(dev, stream, n, k.data_ptr(), board.data_ptr())
    return GameState(
        board=wrap(board),
        step_count=wrap(torch.zeros((n,), dtype=torch.int32, device=rt.device)),
        done=wrap(torch.zeros((n,), dtype=torch.bool, device=rt.device)),
        total_reward=wrap(torch.zeros((n,), dtype=torch.float32, device=rt.device)),
    )


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

    One fused, out-of-place kernel (env2048_step_state_i32_keyed): the input state is only read (functional
    API) and the new state is written to fresh tensors -- no copies on either side.  Finished games are
    frozen exactly like the reference's done-guard (env.py:103-104): state unchanged, reward 0.0, done True,
    step_count not advanced.  Environment i draws its tile from keys[i] alone (vmap semantics).
    """
    rt = runtime.current()
    board_in = _board_tensor(states.board).reshape(-1, 4, 4)
    n = board_in.shape[0]
    k = keys_on_device(keys, rt.device)
    if k.shape[0] != n:
        raise ValueError(f'batched_step: {n} environments but {k.shape[0]} keys')
    steps_in = _dev(states.step_count, torch.int32, n, rt)
    total_in = _dev(states.total_reward, torch.float32, n, rt)
    acts = _dev(actions, torch.int32, n, rt)
    d = to_tensor(states.done)
    if d.dtype == torch.bool and d.device == rt.device:
        done_in = d.reshape(n).contiguous().view(torch.uint8)  # same bytes, no copy
    else:
        done_in = _dev(d.to(torch.uint8), torch.uint8, n, rt)
    board = torch.empty_like(board_in)
    i32 = torch.empty((2, n), dtype=torch.int32, device=rt.device)   # step_count, max_tile
    f32 = torch.empty((2, n), dtype=torch.float32, device=rt.device)  # total_reward, reward
    done_u8 = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    i32_ptr, f32_ptr = i32.data_ptr(), f32.data_ptr()  # (addresses taken on the plain tensors: no __torch_function__ detour)
    rt.lib.step_state_i32_keyed(dev, stream, n, k.data_ptr(), board_in.data_ptr(), board.data_ptr(), acts.data_ptr(),
                                steps_in.data_ptr(), i32_ptr, done_in.data_ptr(), done_u8.data_ptr(),
                                total_in.data_ptr(), f32_ptr, f32_ptr + 4 * n, i32_ptr + 4 * n)
    step_count, max_tile = (wrap(t) for t in i32.unbind(0))
    total, reward = (wrap(t) for t in f32.unbind(0))
    done = wrap(done_u8.view(torch.bool))
    new_state = GameState(board=wrap(board), step_count=step_count, done=done, total_reward=total)
    info = {'max_tile': max_tile, 'step_count': step_count, 'total_reward': total}
    return new_state, reward, done, info


def batched_get_observation(states: GameState) -> torch.Tensor:
    """[ref env.py:214-228] (N, 16) float32 log2/16 observations -- one kernel on the int32 boards."""
    rt = runtime.current()
    boards = _board_tensor(states.board).reshape(-1, 16)
    n = boards.shape[0]
    obs = torch.empty((n, 16), dtype=torch.float32, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.status_i32(dev, stream, n, boards.data_ptr(), obs.data_ptr(), None, None)
    return wrap(obs)


def batched_get_legal_actions(states: GameState) -> torch.Tensor:
    """[ref env.py:231-245] (N, 4) bool legal-action masks -- one kernel, written as bool bytes."""
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


def _one_key(key) -> torch.Tensor:
    """A single PRNG key ((2,) uint32 array-like or a plain integer seed) as a (1, 2) int32 host tensor."""
    k0, k1 = key_words(key)
    return torch.tensor([[k0 - (1 << 32) if k0 >= 1 << 31 else k0, k1 - (1 << 32) if k1 >= 1 << 31 else k1]], dtype=torch.int32)


def reset(key) -> GameState:
    """[ref env.py:55-76] new game: board with two tiles, step_count 0, done False, total_reward 0."""
    return _squeeze(batched_reset(_one_key(key)))


def step(state: GameState, action, key):
    """[ref env.py:79-133] -> (GameState, reward f32, done bool, info dict)."""
    new_state, reward, done, info = batched_step(_unsqueeze(state), [int(action)], _one_key(key))
    return _squeeze(new_state), reward[0], done[0], {k: v[0] for k, v in info.items()}


def get_observation(state: GameState) -> torch.Tensor:
    """[ref env.py:136-151] (16,) float32."""
    return batched_get_observation(_unsqueeze(state))[0]


def get_legal_actions(state: GameState) -> torch.Tensor:
    """[ref env.py:154-169] (4,) bool."""
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
