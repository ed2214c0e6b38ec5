"""
Device-resident trajectory recording: what
``reinforce/training/self_play.py:304-308,346-365`` does with per-environment Python lists and one
device->host copy per step, done with zero copies -- the step kernel writes reward / done / legal /
observation of step t straight into row t of ``[T, N]`` tensors the caller owns (the C-ABI takes
caller-owned pointers, so this needs no kernel support beyond env2048_step itself).
"""

from __future__ import annotations

import torch

from .batch import BatchedEnv2048


class TrajectoryBuffer:
    """``horizon`` steps of ``env`` recorded on device.

    Layout (T = horizon, N = env.num_envs):
        actions (T, N) u8, reward (T, N) i32, done (T, N) u8,
        legal (T+1, N) u8 and obs (T+1, N, 16) of the env's obs_dtype  -- row t describes the board the action of step t
        was chosen on; row T the final boards,
        boards (T+1, N, 16) u8 only with ``record_boards=True`` (one 16 B/board device copy per step).
    """

    def __init__(self, env: BatchedEnv2048, horizon: int, *, record_obs: bool = True, record_legal: bool = True,
                 record_boards: bool = False):
        self.env = env
        self.horizon = int(horizon)
        T, n, dev = self.horizon, env.num_envs, env.device
        self.actions = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        self.reward = torch.zeros((T, n), dtype=torch.int32, device=dev)
        self.done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        self.legal = torch.zeros((T + 1, n), dtype=torch.uint8, device=dev) if record_legal else None
        self.obs = torch.zeros((T + 1, n, 16), dtype=env.obs_dtype, device=dev) if record_obs else None
        self.boards = torch.zeros((T + 1, n, 16), dtype=torch.uint8, device=dev) if record_boards else None
        self.t = 0

    def begin(self) -> None:
        """Describe the CURRENT boards in row 0 (call after env.reset() or whenever a new recording starts)."""
        self.t = 0
        if self.legal is not None or self.obs is not None:
            self.env.describe(self.legal[0] if self.legal is not None else None, self.obs[0] if self.obs is not None else None)
        if self.boards is not None:
            self.boards[0].copy_(self.env.boards)

    def step(self, actions: torch.Tensor | None = None):
        """Step the env, recording row t. ``actions`` may be ``self.actions[t]`` itself (write the policy's
        choice there and pass it: no copy) or any (N,) tensor (copied in); ``None`` = on-device random
        actions (then row t of ``actions`` is left untouched: they are a function of (seed, step))."""
        t = self.t
        if t >= self.horizon:
            raise IndexError(f'trajectory buffer full ({self.horizon} steps)')
        if actions is not None:
            row = self.actions[t]
            if actions.data_ptr() != row.data_ptr():
                row.copy_(actions.to(device=row.device, dtype=torch.uint8).reshape(row.shape))
            actions = row
        out = self.env.step(actions, out_reward=self.reward[t], out_done=self.done[t],
                            out_legal=self.legal[t + 1] if self.legal is not None else None,
                            out_obs=self.obs[t + 1] if self.obs is not None else None)
        if self.boards is not None:
            self.boards[t + 1].copy_(self.env.boards)
        self.t = t + 1
        return out

    def returns(self) -> torch.Tensor:
        """Undiscounted sum of rewards per environment over the recorded steps (int64)."""
        return self.reward[: self.t].to(torch.int64).sum(dim=0)
