"""
``BatchedEnvNxN`` -- N independent 2048 boards of any size S x S (2 <= S <= 8) stepped in lock-step on one GPU: the batch face
of the general-size kernels (``env2048_nxn_*``, csrc/nxn_kernels.hip) that the ``TwentyFortyEight(size=S)`` mirror runs one
board at a time [ref: twentyfortyeight/envs/twentyfortyeight.py:21,84-133; the reference's own timing script runs sizes 4, 6
and 8, tests/test_perf_utils.py:47].  Same contract as ``BatchedEnv2048`` where they overlap: boards are uint8 log2 exponents
(one byte per cell, row-major), every environment draws from the Philox stream of its GLOBAL index ``env_base + i`` (a batch
sliced over ranks plays the same games as the whole), ``step(None)`` draws the moves in the kernel, ``auto_reset`` restarts a
finished game inside the launch that finishes it.  For S = 4 it plays exactly the games ``BatchedEnv2048`` plays (the packed
kernels are the throughput path for that size).  No CPU fallback: a HIP device or an injected library (tests).
"""

from __future__ import annotations

import torch

from . import capi
from .batch import _host_copy, _p, _stream_handle


class BatchedEnvNxN:
    def __init__(self, num_envs: int, size: int, seed: int = 0, *, device='cuda', env_base: int = 0, auto_reset: bool = False,
                 with_legal: bool = False, lib=None):
        if not isinstance(size, int) or not 2 <= size <= capi.NXN_MAX_SIZE:
            raise ValueError(f'BatchedEnvNxN: size must be 2 ... {capi.NXN_MAX_SIZE}, got {size!r}')
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != 'cuda':
                raise RuntimeError(f'BatchedEnvNxN runs on HIP devices only (got device={self.device}); there is no CPU fallback')
            lib = capi.library()
            if lib.device_count() <= 0:
                raise RuntimeError('no HIP device visible to libenv2048')
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.lib = lib
        self.num_envs, self.size = int(num_envs), int(size)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env_base = int(env_base)
        self.auto_reset = bool(auto_reset)
        self.step_index = 0
        n = self.num_envs
        self.boards = torch.zeros((n, size, size), dtype=torch.uint8, device=self.device)
        self.reward = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self.done = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        self.legal = torch.zeros((n,), dtype=torch.uint8, device=self.device) if with_legal else None
        self._actions_u8 = None

    def _actions(self, actions: torch.Tensor) -> torch.Tensor:
        """The caller's moves as the kernels read them: a contiguous uint8 tensor of shape (num_envs,) on this device.  Anything else
        with the right number of elements (an int64 argmax, a host tensor, a strided view) is converted into a persistent uint8
        buffer -- one copy kernel, no allocation -- as ``BatchedEnv2048.step`` does; a wrong count raises."""
        if not isinstance(actions, torch.Tensor):
            actions = torch.as_tensor(actions)
        if actions.numel() != self.num_envs:
            raise ValueError(f'expected {self.num_envs} actions, got {actions.numel()}')
        if actions.dtype == torch.uint8 and actions.device == self.device and actions.dim() == 1 and actions.is_contiguous():
            return actions
        if actions.dtype.is_floating_point or actions.dtype.is_complex or actions.dtype == torch.bool:
            raise ValueError(f'actions must be integers (0 left, 1 up, 2 right, 3 down), got {actions.dtype}')
        staged = self._actions_u8
        if staged is None:
            staged = self._actions_u8 = torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device)
        staged.copy_(actions.reshape(self.num_envs))  # wraps mod 256, the kernel reads mod 4
        return staged

    def reset(self) -> torch.Tensor:
        """Two tiles on every (empty) board, from the reset block of the current step index.  Returns the boards (in place)."""
        dev, stream = _stream_handle(self.device)
        self.lib.nxn_reset(dev, stream, self.num_envs, self.size, self.env_base, _p(self.boards), _p(self.done), self.seed, self.step_index)
        self.step_index += 1
        if self.legal is not None:
            self.lib.nxn_status(dev, stream, self.num_envs, self.size, _p(self.boards), None, _p(self.legal), None, None)
        return self.boards

    def step(self, actions: torch.Tensor | None = None):
        """One move per board (actions 0 left, 1 up, 2 right, 3 down, taken mod 4; a contiguous uint8 tensor on this device is read
        in place, anything else goes through ``_actions``; None: drawn from the step word in the kernel) -> (boards, reward int32,
        done uint8), all updated in place; ``legal`` describes the boards left behind."""
        flags = capi.FLAG_AUTO_RESET if self.auto_reset else 0
        if actions is None:
            flags |= capi.FLAG_RANDOM_ACTIONS
        else:
            actions = self._actions(actions)
        dev, stream = _stream_handle(self.device)
        self.lib.nxn_step(dev, stream, self.num_envs, self.size, self.env_base, _p(self.boards), _p(actions), _p(self.reward), _p(self.done),
                          _p(self.legal), self.seed, self.step_index, flags)
        self.step_index += 1
        return self.boards, self.reward, self.done

    def status(self) -> dict:
        """done / legal bitmask / number of empty cells / log2(max tile) of the current boards (one launch)."""
        n = self.num_envs
        out = {k: torch.empty((n,), dtype=torch.uint8, device=self.device) for k in ('done', 'legal', 'n_empty', 'max_exp')}
        dev, stream = _stream_handle(self.device)
        self.lib.nxn_status(dev, stream, n, self.size, _p(self.boards), _p(out['done']), _p(out['legal']), _p(out['n_empty']), _p(out['max_exp']))
        return out

    def afterstates(self, actions: torch.Tensor):
        """latent_state of every board for its action, no spawn -> (boards (N, S, S), reward int32, changed uint8); the env is untouched."""
        n = self.num_envs
        actions = self._actions(actions)
        out = torch.empty_like(self.boards)
        reward = torch.empty((n,), dtype=torch.int32, device=self.device)
        changed = torch.empty((n,), dtype=torch.uint8, device=self.device)
        dev, stream = _stream_handle(self.device)
        self.lib.nxn_afterstate(dev, stream, n, self.size, _p(self.boards), _p(actions), _p(out), _p(reward), _p(changed))
        return out, reward, changed

    def values(self, dtype=torch.int64) -> torch.Tensor:
        """The boards as tile values (2^e, 0 for empty)."""
        e = self.boards.to(torch.int64)
        return torch.where(e > 0, torch.ones_like(e) << e, torch.zeros_like(e)).to(dtype)

    # -- checkpoint / resume (the layout of BatchedEnv2048.state_dict plus the board size) -----------------------------------------
    def state_dict(self) -> dict:
        """Everything that determines the future of this batch, as host tensors / integers (``torch.save`` -able): stepping a
        restored batch gives byte for byte what the original would have produced."""
        return {'format': 1, 'size': self.size, 'num_envs': self.num_envs, 'seed': self.seed, 'env_base': self.env_base,
                'step_index': int(self.step_index), 'auto_reset': self.auto_reset, 'boards': _host_copy(self.boards),
                'reward': _host_copy(self.reward), 'done': _host_copy(self.done)}

    def load_state_dict(self, state: dict) -> None:
        """Restore ``state_dict()`` output into this batch (same size and number of environments; seed, index range, step index
        and the auto-reset policy come from the checkpoint); ``legal`` is recomputed for the restored boards."""
        if state.get('format') != 1 or int(state['size']) != self.size or int(state['num_envs']) != self.num_envs:
            raise ValueError(f'checkpoint of {state.get("num_envs")} boards of size {state.get("size")} (format {state.get("format")}) '
                             f'does not fit this batch of {self.num_envs} boards of size {self.size}')
        shape = (self.num_envs, self.size, self.size)
        self.boards.copy_(torch.as_tensor(state['boards'], dtype=torch.uint8).reshape(shape))
        self.reward.copy_(torch.as_tensor(state['reward'], dtype=torch.int32).reshape(self.num_envs))
        self.done.copy_(torch.as_tensor(state['done'], dtype=torch.uint8).reshape(self.num_envs))
        self.seed = int(state['seed']) & 0xFFFFFFFFFFFFFFFF
        self.env_base, self.step_index = int(state['env_base']), int(state['step_index'])
        self.auto_reset = bool(state['auto_reset'])
        if self.legal is not None:
            dev, stream = _stream_handle(self.device)
            self.lib.nxn_status(dev, stream, self.num_envs, self.size, _p(self.boards), None, _p(self.legal), None, None)


__all__ = ['BatchedEnvNxN']
