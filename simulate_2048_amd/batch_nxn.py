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
from .batch import _p, _stream_handle


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

    def reset(self) -> torch.Tensor:
        """Two tiles on every (empty) board, from the reset block of the current step index.  Returns the boards (in place)."""
        dev, stream = _stream_handle(self.device)
        self.lib.nxn_reset(dev, stream, self.num_envs, self.size, self.env_base, _p(self.boards), _p(self.done), self.seed, self.step_index)
        self.step_index += 1
        if self.legal is not None:
            self.lib.nxn_status(dev, stream, self.num_envs, self.size, _p(self.boards), None, _p(self.legal), None, None)
        return self.boards

    def step(self, actions: torch.Tensor | None = None):
        """One move per board (uint8 actions 0 left, 1 up, 2 right, 3 down on this device; None: drawn from the step word in the
        kernel) -> (boards, reward int32, done uint8), all updated in place; ``legal`` describes the boards left behind."""
        flags = capi.FLAG_AUTO_RESET if self.auto_reset else 0
        if actions is None:
            flags |= capi.FLAG_RANDOM_ACTIONS
        else:
            if actions.dtype != torch.uint8 or actions.device != self.device or actions.shape != (self.num_envs,) or not actions.is_contiguous():
                raise ValueError(f'actions must be a contiguous uint8 tensor of shape ({self.num_envs},) on {self.device}')
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

    def state_dict(self) -> dict:
        return {'boards': self.boards.clone(), 'step_index': self.step_index, 'seed': self.seed, 'env_base': self.env_base, 'size': self.size}

    def load_state_dict(self, state: dict) -> None:
        if int(state['size']) != self.size or tuple(state['boards'].shape) != tuple(self.boards.shape):
            raise ValueError('state_dict of another shape')
        self.boards.copy_(state['boards'])
        self.step_index, self.seed, self.env_base = int(state['step_index']), int(state['seed']), int(state['env_base'])


__all__ = ['BatchedEnvNxN']
