"""
Device-resident trajectory recording: what
``reinforce/training/self_play.py:304-308,346-392`` does with per-game Python lists, one device->host copy per
array per move and an O(N) Python loop per move -- done with zero copies: the step kernel writes action / reward /
done / legal / observation / max-tile of step t straight into row t of ``[T, N]`` tensors (the C-ABI takes
caller-owned pointers), the policy's outputs (search policy, search value) go into rows of the same shape, and one
pair of kernels (env2048_count_episodes / env2048_segment_episodes) cuts the rows into per-game records carrying what
``Trajectory(...)`` carries (:381-390): rows, done, total_reward, max_tile.

Two recording modes:
* plain (``ring=False``): ``horizon`` steps, then full -- one self-play call of the reference.
* ring  (``ring=True``):  step t lands in row t % horizon, ``generation`` counts wrap-arounds; the window that can be
  cut into episodes is the last ``horizon - 1`` steps (the observation row of the oldest step has been reused).

Without auto-reset (the reference's loop) a finished game stays finished: its rows after the first ``done`` belong to
no episode ("store step data only for active games", :359-368).  With ``auto_reset`` every environment plays game
after game and every game is a record.
"""

from __future__ import annotations

from typing import NamedTuple

import numpy as np
import torch

from . import capi
from .batch import BatchedEnv2048, _p, _stream_handle


class Trajectory(NamedTuple):
    """One finished (or cut) game in the field layout of the reference's ``Trajectory``
    (reinforce/training/replay_buffer.py:35-66): host NumPy arrays, ready for ``ReplayBuffer.add``."""

    observations: np.ndarray  # (T, 16) float32
    actions: np.ndarray       # (T,) int32
    rewards: np.ndarray       # (T,) float32
    policies: np.ndarray      # (T, A) float32 (zeros when no policy rows were recorded)
    values: np.ndarray        # (T,) float32 (zeros when no value rows were recorded)
    done: bool = True
    total_reward: float = 0.0
    max_tile: int = 0

    def __len__(self) -> int:
        return len(self.observations)


class _Rows:
    """Rows of a (rows, ...) tensor: ``rows[r]`` is the view of row r (all made at once for buffers of up to 4096 rows -- one
    unbind --, on first use beyond that), ``rows.address[r]`` its device address (arithmetic, no tensor involved)."""

    __slots__ = ('tensor', 'views', 'address')

    class _Addresses:
        __slots__ = ('base', 'pitch')

        def __init__(self, tensor: torch.Tensor):
            self.base, self.pitch = tensor.data_ptr(), tensor.stride(0) * tensor.element_size()

        def __getitem__(self, row: int) -> int:
            return self.base + row * self.pitch

    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor
        self.views = list(tensor.unbind(0)) if tensor.shape[0] <= 4096 else [None] * tensor.shape[0]
        self.address = _Rows._Addresses(tensor)

    def __getitem__(self, row: int) -> torch.Tensor:
        view = self.views[row]
        if view is None:
            view = self.views[row] = self.tensor[row]
        return view


class TrajectoryBuffer:
    """Rows of ``env`` recorded on device.

    Layout (C = horizon, N = env.num_envs; step t uses physical row t in plain mode, t % C in ring mode):
        actions (C, N) u8, reward (C, N) i32 -- or f32 log2(score)/16 with ``normalize_reward=True`` --, done (C, N) u8,
        max_exp (C, N) u8   log2 of the max tile after the move, before an auto-reset (``record_max_tile``),
        policy (C, N, A) f32 and value (C, N) f32   the caller's search outputs (``record_policy`` / ``record_value``),
        legal (C+1, N) u8 and obs (C+1, N, 16) of the env's obs_dtype -- row t describes the board the action of step t was
        chosen on, row t+1 the board after it (ring mode: C rows, row (t+1) % C),
        boards (C+1, N, 16) u8 only with ``record_boards=True`` (one 16 B/board device copy per step).
    """

    def __init__(self, env: BatchedEnv2048, horizon: int, *, record_obs: bool = True, record_legal: bool = True,
                 record_boards: bool = False, record_policy: bool = False, record_value: bool = False, num_actions: int = 4,
                 record_max_tile: bool = True, normalize_reward: bool = False, ring: bool = False):
        self.env = env
        self.horizon = int(horizon)
        if self.horizon < 2:
            raise ValueError('horizon must be at least 2')
        self.ring = bool(ring)
        self.normalize_reward = bool(normalize_reward)
        C, n, dev = self.horizon, env.num_envs, env.device
        extra = 0 if self.ring else 1
        self.actions = torch.zeros((C, n), dtype=torch.uint8, device=dev)
        self.reward = torch.zeros((C, n), dtype=torch.float32 if self.normalize_reward else torch.int32, device=dev)
        self.done = torch.zeros((C, n), dtype=torch.uint8, device=dev)
        self.max_exp = torch.zeros((C, n), dtype=torch.uint8, device=dev) if record_max_tile else None
        self.policy = torch.zeros((C, n, num_actions), dtype=torch.float32, device=dev) if record_policy else None
        self.value = torch.zeros((C, n), dtype=torch.float32, device=dev) if record_value else None
        self.legal = torch.zeros((C + extra, n), dtype=torch.uint8, device=dev) if record_legal else None
        self.obs = torch.zeros((C + extra, n, 16), dtype=env.obs_dtype, device=dev) if record_obs else None
        self.boards = torch.zeros((C + extra, n, 16), dtype=torch.uint8, device=dev) if record_boards else None
        self.t = 0  # steps recorded since begin()
        self._unrecorded_random = False
        # Row addresses by arithmetic and row views made once each: a recorded move then does no tensor indexing (six index
        # operations cost ~10 us of host time per move -- more than the move at small batches).
        names = ('actions', 'reward', 'done', 'max_exp', 'legal', 'obs', 'policy', 'value', 'boards')
        self._rows = {name: None if getattr(self, name) is None else _Rows(getattr(self, name)) for name in names}
        self._ptrs = {name: None if rows is None else rows.address for name, rows in self._rows.items()}

    # -- bookkeeping --------------------------------------------------------------------------------------------
    @property
    def generation(self) -> int:
        """Ring mode: how many times the rows have wrapped around."""
        return self.t // self.horizon if self.ring else 0

    def _row(self, t: int) -> int:
        return t % self.horizon if self.ring else t

    def window(self) -> tuple[int, int]:
        """(first step, number of steps) that are completely present in the rows right now."""
        if not self.ring:
            return 0, self.t
        keep = self.horizon - 1 if (self.obs is not None or self.legal is not None or self.boards is not None) else self.horizon
        count = min(self.t, keep)
        return self.t - count, count

    def begin(self) -> None:
        """Describe the CURRENT boards in row 0 (call after env.reset() or whenever a new recording starts)."""
        self.t = 0
        self._unrecorded_random = False
        if self.legal is not None or self.obs is not None:
            self.env.describe(self.legal[0] if self.legal is not None else None, self.obs[0] if self.obs is not None else None)
        if self.boards is not None:
            self.boards[0].copy_(self.env.boards)

    def step(self, actions: torch.Tensor | None = None, *, policy: torch.Tensor | None = None, value: torch.Tensor | None = None):
        """Step the env, recording row t. ``actions`` may be ``self.actions[row]`` itself (write the policy's
        choice there and pass it: no copy) or any (N,) tensor (copied in); ``None`` = on-device random actions, which
        are written into the action row first (env2048_random_actions) so that the recording says what was played.
        ``policy`` (N, A) / ``value`` (N,): the search outputs of this move (self_play.py:339-340), copied into their rows;
        or write ``self.policy[row]`` / ``self.value[row]`` directly before calling."""
        t = self.t
        if not self.ring and t >= self.horizon:
            raise IndexError(f'trajectory buffer full ({self.horizon} steps)')
        row, nxt = self._row(t), self._row(t + 1)
        rows, ptrs, env = self._rows, self._ptrs, self.env
        if env.step_counter is not None:
            return self._step_with_device_counter(actions, policy, value, row, nxt)
        action_row = rows['actions'][row]
        if actions is not None:
            if actions.data_ptr() != ptrs['actions'][row]:
                action_row.copy_(actions.to(device=action_row.device, dtype=torch.uint8).reshape(action_row.shape))
        else:
            # on-device random play: write the moves the step WOULD draw into the action row and play them from there -- the
            # same step (same words), and the row says what was played (these rows used to stay zero)
            env.random_actions(action_row)
        if policy is not None:
            if self.policy is None:
                raise ValueError('this buffer was built without record_policy=True')
            if policy.data_ptr() != ptrs['policy'][row]:
                rows['policy'][row].copy_(policy)
        if value is not None:
            if self.value is None:
                raise ValueError('this buffer was built without record_value=True')
            if value.data_ptr() != ptrs['value'][row]:
                rows['value'][row].copy_(value)
        env._record_step(ptrs['actions'][row], ptrs['reward'][row], ptrs['done'][row], None if self.legal is None else ptrs['legal'][nxt],
                         None if self.obs is None else ptrs['obs'][nxt], None if self.max_exp is None else ptrs['max_exp'][row],
                         capi.FLAG_REWARD_LOG2_F32 if self.normalize_reward else 0)
        if self.boards is not None:
            rows['boards'][nxt].copy_(env.boards)
        self.t = t + 1
        return env.boards, rows['reward'][row], rows['done'][row]

    def _step_with_device_counter(self, actions, policy, value, row, nxt):
        """step() for an env whose step index lives on the device (graph capture): the general BatchedEnv2048.step path."""
        if actions is not None:
            dst = self.actions[row]
            if actions.data_ptr() != dst.data_ptr():
                dst.copy_(actions.to(device=dst.device, dtype=torch.uint8).reshape(dst.shape))
            actions = dst
        else:
            self._unrecorded_random = True  # (step index on the device: the moves cannot be materialised beforehand)
        if policy is not None:
            if self.policy is None:
                raise ValueError('this buffer was built without record_policy=True')
            if policy.data_ptr() != self.policy[row].data_ptr():
                self.policy[row].copy_(policy)
        if value is not None:
            if self.value is None:
                raise ValueError('this buffer was built without record_value=True')
            if value.data_ptr() != self.value[row].data_ptr():
                self.value[row].copy_(value)
        out = self.env.step(actions, out_reward=self.reward[row], out_done=self.done[row],
                            out_legal=self.legal[nxt] if self.legal is not None else None,
                            out_obs=self.obs[nxt] if self.obs is not None else None,
                            out_max_exp=self.max_exp[row] if self.max_exp is not None else None,
                            normalized_reward=self.normalize_reward)
        if self.boards is not None:
            self.boards[nxt].copy_(self.env.boards)
        self.t += 1
        return out

    def _require_action_rows(self) -> None:
        if self._unrecorded_random:
            raise RuntimeError('this recording holds random-action steps of an env with a device-side step counter: their moves were '
                               'not written to the action rows, so trajectories cannot be exported (pass explicit actions, or build '
                               'the env without device_step_counter)')

    def returns(self) -> torch.Tensor:
        """Undiscounted sum of the recorded reward rows per environment (int64, or float32 sums when normalised)."""
        first, count = self.window()
        rows = self.reward[: self.t] if not self.ring else self.reward[[self._row(first + r) for r in range(count)]]
        return rows.sum(dim=0) if self.normalize_reward else rows.to(torch.int64).sum(dim=0)

    # -- episodes -------------------------------------------------------------------------------------------------
    def episodes(self, *, frozen: bool | None = None, want_episode_of: bool = False) -> dict:
        """Cut the recorded window into per-game records on device (self_play.py:372-392 without the Python loop).

        ``frozen`` (default: ``not env.auto_reset``): the rows were recorded without auto-reset, so a game has ONE episode
        and its rows after the first ``done`` are not part of it (the reference's ``active_mask``).
        Returns tensors of equal length E, ordered by environment, then by start:
            env (int32), start (int64, ABSOLUTE step index since begin()), length (int32), terminated (bool),
            max_tile (int64 tile VALUE; 0 if max tiles were not recorded), total_reward (float64),
        plus ``first_step`` / ``num_steps`` of the window, and with ``want_episode_of`` an int32 (C, N) tensor giving, for
        every physical row, the index of the record it belongs to (-1: none; rows outside the window are not touched)."""
        env = self.env
        frozen = (not env.auto_reset) if frozen is None else bool(frozen)
        first, count = self.window()
        n, C = env.num_envs, self.horizon
        flags = (capi.EPISODES_FROZEN if frozen else 0) | (capi.EPISODES_REWARD_F32 if self.normalize_reward else 0)
        dev, stream = _stream_handle(env.device)
        counts = torch.zeros((n,), dtype=torch.int32, device=env.device)
        first_row = self._row(first) if count else 0
        env.lib.count_episodes(dev, stream, n, n, count, first_row, C, _p(self.done), flags, _p(counts))
        ends = torch.cumsum(counts.to(torch.int64), dim=0)
        offsets = (ends - counts).contiguous()
        total = int(ends[-1]) if n else 0
        raw = torch.zeros((max(total, 1), 24), dtype=torch.uint8, device=env.device)
        episode_of = torch.full((C, n), -1, dtype=torch.int32, device=env.device) if want_episode_of else None
        env.lib.segment_episodes(dev, stream, n, n, count, first_row, C, _p(self.done), _p(self.reward), _p(self.max_exp), flags,
                                 _p(offsets), _p(raw), _p(episode_of))
        raw = raw[:total]
        words = raw.view(torch.int32)  # (E, 6): env, start, length, {terminated, max_exp, reserved}, total_reward (2 words)
        packed = words[:, 3]
        max_exp = (packed >> 8) & 0xFF
        out = {
            'env': words[:, 0],
            'start': words[:, 1].to(torch.int64) + first,
            'length': words[:, 2],
            'terminated': (packed & 0xFF) != 0,
            'max_tile': torch.where(max_exp > 0, torch.ones_like(max_exp, dtype=torch.int64) << max_exp.to(torch.int64), 0),
            'total_reward': raw.view(torch.float64)[:, 2],
            'first_step': first,
            'num_steps': count,
        }
        if want_episode_of:
            out['episode_of'] = episode_of
        return out

    def trajectory(self, episodes: dict, k: int) -> dict:
        """The rows of episode record ``k`` gathered into the fields of the reference's ``Trajectory``
        (reinforce/training/replay_buffer.py:35-66): observations (L, 16), actions (L,), rewards (L,), policies (L, A),
        values (L,), done, total_reward, max_tile -- device tensors / Python scalars."""
        self._require_action_rows()
        e = int(episodes['env'][k])
        start, length = int(episodes['start'][k]), int(episodes['length'][k])
        rows = torch.as_tensor([self._row(start + r) for r in range(length)], dtype=torch.int64, device=self.env.device)
        pick = lambda t: None if t is None else t[rows, e]
        return {
            'observations': pick(self.obs),
            'actions': pick(self.actions),
            'rewards': pick(self.reward),
            'policies': pick(self.policy),
            'values': pick(self.value),
            'done': bool(episodes['terminated'][k]),
            'total_reward': float(episodes['total_reward'][k]),
            'max_tile': int(episodes['max_tile'][k]),
        }

    def export_trajectories(self, *, frozen: bool | None = None, min_length: int = 1) -> list[Trajectory]:
        """Every recorded game as a host-side ``Trajectory`` -- what ``play_parallel_games`` returns
        (reinforce/training/self_play.py:372-392) -- with ONE device-to-host copy per row array instead of one per move:
        the episode records are cut on device (``episodes``), the rows travel once, and the per-game arrays are NumPy
        slices of them.  Needs observation rows (``record_obs=True``)."""
        if self.obs is None:
            raise ValueError('export_trajectories needs observation rows (record_obs=True)')
        self._require_action_rows()
        ep = self.episodes(frozen=frozen)
        first, count = ep['first_step'], ep['num_steps']
        order = [self._row(first + r) for r in range(count)]  # logical step -> physical row, once
        rows = torch.as_tensor(order, dtype=torch.int64, device=self.env.device)
        obs = self.obs[rows].to(torch.float32).cpu().numpy()           # (count, N, 16)
        actions = self.actions[rows].cpu().numpy().astype(np.int32)    # (count, N)
        rewards = self.reward[rows].to(torch.float32).cpu().numpy()
        num_actions = self.policy.shape[-1] if self.policy is not None else 4
        policies = self.policy[rows].cpu().numpy() if self.policy is not None else None
        values = self.value[rows].cpu().numpy() if self.value is not None else None
        env_idx = ep['env'].cpu().numpy()
        start = (ep['start'].cpu().numpy() - first).astype(np.int64)
        length = ep['length'].cpu().numpy().astype(np.int64)
        terminated = ep['terminated'].cpu().numpy()
        total = ep['total_reward'].cpu().numpy()
        max_tile = ep['max_tile'].cpu().numpy()
        out = []
        for k in range(len(env_idx)):
            e, s, n = int(env_idx[k]), int(start[k]), int(length[k])
            if n < min_length:
                continue
            out.append(Trajectory(
                observations=obs[s:s + n, e], actions=actions[s:s + n, e], rewards=rewards[s:s + n, e],
                policies=policies[s:s + n, e] if policies is not None else np.zeros((n, num_actions), dtype=np.float32),
                values=values[s:s + n, e] if values is not None else np.zeros((n,), dtype=np.float32),
                done=bool(terminated[k]), total_reward=float(total[k]), max_tile=int(max_tile[k])))
        return out
