"""
Host-side batched 2048 environment: owns the device buffers (torch tensors are used purely as
device-memory handles) and drives the HIP kernels through the C-ABI (``capi.Library``).

State of a batch = ``boards`` (N x 16 packed log2 bytes) + ``(seed, step_index)``; everything else
is an output buffer.  ``env_base`` is the global index of element 0, so a batch that is sliced
across GPUs (``sharding.shard_range``) produces exactly the boards the unsliced batch would.

Mirrors the batched half of the reference API (``reinforce/game/env.py:176-245`` batched_reset /
batched_step / batched_get_observation / batched_get_legal_actions) in packed form; the literal
drop-in shims built on top of it live in ``simulate_2048_amd.reinforce.game``.
"""

from __future__ import annotations

import torch

from . import capi


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)  # int handle without building a Stream object


def _stream_handle(device: torch.device):
    """(device ordinal, hipStream_t) of torch's CURRENT stream on `device` (whose index is fixed at construction)."""
    if device.type != 'cuda':
        return -1, None
    index = device.index
    if _raw_stream is not None:
        return index, _raw_stream(index)
    return index, torch.cuda.current_stream(device).cuda_stream


def _host_copy(t: torch.Tensor) -> torch.Tensor:
    """A host tensor that shares nothing with `t` (``.cpu()`` of a host tensor -- the injected-library tests -- is the tensor itself)."""
    return t.detach().to('cpu', copy=True)


def _p(t):
    return None if t is None else t.data_ptr()


class BatchedEnv2048:
    """N independent 2048 boards stepped in lock-step on one GPU.

    Parameters
    ----------
    num_envs : int
        Boards in this batch (this rank's slice when sharded).
    seed : int
        64-bit Philox key shared by the whole (global) batch.
    device : str | torch.device
        A HIP device ('cuda', 'cuda:0', ...).  CPU devices are refused unless a non-default
        ``lib`` is injected (the test-suite does that with an oracle-backed stand-in).
    env_base : int
        Global index of board 0 (multi-GPU slicing).
    auto_reset : bool
        Replace a board by a fresh one in the same step in which it became terminal
        (new capability; the reference freezes finished games, reinforce/game/env.py:103-104).
    with_legal, with_obs : bool
        Also emit the legal-action bitmask / the observation from the step kernel
        (what reinforce/training/self_play.py:316-317 asks the env for every move).
    device_step_counter : bool
        Keep the step index in device memory (``step_counter``, one int64) and let every ``step`` launch take its index
        as counter + offset.  A ``torch.cuda.graph`` that captured ``step`` calls followed by ``commit_steps()`` can then be
        REPLAYED -- each replay continues the random stream -- which takes the host out of a launch-bound per-move loop
        (2-3 us per step as a graph instead of 3-4 us of launch cost at small batches).  ``step_index`` then reads the
        counter back (a synchronisation).
    obs_dtype : torch.float32 | torch.float16 | torch.bfloat16
        Element type of ``obs``.  The values (log2(tile)/16) are exact in all three; the 16-bit forms halve
        the output that bounds the step-with-observations path and are what a bf16/fp16 network consumes.
    """

    _OBS_FLAGS = {torch.float32: 0, torch.float16: capi.FLAG_OBS_F16, torch.bfloat16: capi.FLAG_OBS_BF16}


    def __init__(self, num_envs: int, seed: int = 0, *, device='cuda', env_base: int = 0, auto_reset: bool = False,
                 with_legal: bool = False, with_obs: bool = False, obs_dtype=torch.float32, device_step_counter: bool = False,
                 lib=None):
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != 'cuda':
                raise RuntimeError(
                    f'BatchedEnv2048 runs on HIP devices only (got device={self.device}); there is no CPU fallback'
                )
            lib = capi.library()
            if lib.device_count() <= 0:
                raise RuntimeError('no HIP device visible to libenv2048')
        if self.device.type == 'cuda' and self.device.index is None:
            # resolve 'cuda' ONCE: the buffers live on the device that is current now, and every later launch must
            # go to that device and one of its streams even if the process switches devices in between
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.lib = lib
        self.num_envs = int(num_envs)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env_base = int(env_base)
        self.auto_reset = bool(auto_reset)
        self._step_index = 0
        n = self.num_envs
        self.step_counter = torch.zeros((1,), dtype=torch.int64, device=self.device) if device_step_counter else None
        self._pending = 0  # steps enqueued since the device counter was last advanced (their offsets)
        self._in_capture = False
        self._record_args = None  # (see _record_step)
        self._actions_u8 = None  # (see step: actions of another dtype / device are converted into this buffer)
        self.boards = torch.zeros((n, 16), dtype=torch.uint8, device=self.device)
        self.reward = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self.done = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        self.legal = torch.zeros((n,), dtype=torch.uint8, device=self.device) if with_legal else None
        if obs_dtype not in self._OBS_FLAGS:
            raise ValueError(f'obs_dtype must be float32, float16 or bfloat16, got {obs_dtype}')
        self.obs_dtype = obs_dtype
        self.obs = torch.zeros((n, 16), dtype=obs_dtype, device=self.device) if with_obs else None
        # the persistent buffers never move: keep their addresses (saves ~6 data_ptr() calls per step)
        self._ptr_boards, self._ptr_reward, self._ptr_done = _p(self.boards), _p(self.reward), _p(self.done)
        self._ptr_legal, self._ptr_obs = _p(self.legal), _p(self.obs)

    # ------------------------------------------------------------------------------------------
    @property
    def step_index(self) -> int:
        """Index of the NEXT step (with ``device_step_counter``: read back from the device)."""
        if self.step_counter is not None:
            return int(self.step_counter[0]) + self._pending
        return self._step_index

    @step_index.setter
    def step_index(self, value: int) -> None:
        self._step_index = int(value)
        if self.step_counter is not None:
            self.step_counter[0] = int(value)
            self._pending = 0

    def commit_steps(self) -> None:
        """``device_step_counter`` only: advance the device counter by the steps enqueued since the last commit (one
        one-thread kernel on the current stream).  Call it at the END of a region captured into a graph, so that every
        replay of the graph plays the NEXT steps.  Eager steps commit by themselves; a capture that ends without it is reported
        by the next step() / commit_steps()."""
        if self.step_counter is None or self._pending == 0:
            return
        if self._in_capture and not (self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            pending, self._pending, self._in_capture = self._pending, 0, False  # (committing after the capture block: too late)
            raise RuntimeError(f'commit_steps() must be called INSIDE the captured region ({pending} captured step(s) were left '
                               'uncommitted: replays of that graph repeat the same step indices)')
        dev, stream = _stream_handle(self.device)
        self.lib.advance_counter(dev, stream, _p(self.step_counter), self._pending)
        self._pending = 0

    @property
    def flags(self) -> int:
        return (capi.FLAG_AUTO_RESET if self.auto_reset else 0) | self._OBS_FLAGS[self.obs_dtype]

    def reset(self, step_index: int | None = None) -> torch.Tensor:
        """Fresh boards for the whole batch (two spawned tiles each). Returns ``boards``."""
        if step_index is not None:
            self.step_index = int(step_index)
        dev, stream = _stream_handle(self.device)
        self.lib.reset(dev, stream, self.num_envs, self.env_base, _p(self.boards), _p(self.done), self.seed,
                       self.step_index)
        self.reward.zero_()
        self.reset_episode_carry()  # every environment starts a new game: nothing is "in progress" any more
        if self.legal is not None or self.obs is not None:
            self.describe(self.legal, self.obs)
        return self.boards

    def reset_episode_carry(self) -> None:
        """Forget the game every environment is IN for ``rollout_episodes`` (running length / return, and the "this game was
        already counted" marks of non-auto-reset runs).  ``reset()`` and ``set_boards()`` call it; call it yourself after
        advancing the boards by other means (``step`` / ``rollout`` / ``replay`` do not see the carry) if the next
        ``rollout_episodes`` should count games from the current boards on."""
        if getattr(self, '_ep_length', None) is not None:
            self._ep_length.zero_()
            self._ep_return.zero_()

    def describe(self, legal: torch.Tensor | None, obs: torch.Tensor | None) -> None:
        """Legal mask / observation of the CURRENT boards into the given tensors (obs of this env's obs_dtype;
        the status kernel writes f32, a 16-bit obs is converted from it -- exact)."""
        dev, stream = _stream_handle(self.device)
        if obs is None or obs.dtype == torch.float32:
            self.lib.status(dev, stream, self.num_envs, _p(self.boards), None, _p(legal), _p(obs), None, None)
            return
        tmp = torch.empty((self.num_envs, 16), dtype=torch.float32, device=self.device)
        self.lib.status(dev, stream, self.num_envs, _p(self.boards), None, _p(legal), _p(tmp), None, None)
        obs.copy_(tmp)

    def step(self, actions: torch.Tensor | None = None, *, out_reward=None, out_done=None, out_legal=None, out_obs=None,
             out_max_exp=None, normalized_reward: bool = False):
        """One transition per board, in place; asynchronous on torch's current stream.

        ``actions``: uint8 tensor (N,) on the same device, values taken mod 4 (0 left, 1 up,
        2 right, 3 down); ``None`` draws uniform random actions on device (low two bits of each step word).
        ``out_*``: optional caller-owned tensors the kernel writes INSTEAD of the persistent buffers
        (e.g. slices of a ``[T, N]`` trajectory buffer -- see ``trajectory.TrajectoryBuffer``).
        ``out_max_exp``: optional uint8 (N,) tensor receiving log2(max tile) of every board after the move and BEFORE an
        auto-reset replaces it (the max tile of a game that ends in this step).
        ``normalized_reward``: write the log-normalised reward log2(score)/16 as float32 (``normalize_reward`` of the
        reference, twentyfortyeight/utils/normalize.py:6-33) -- ``out_reward`` must then be a float32 tensor.
        Returns ``(boards, reward, done)`` -- views of the buffers that were written.
        """
        flags = self.flags
        if actions is None:
            flags |= capi.FLAG_RANDOM_ACTIONS
        else:
            if actions.numel() != self.num_envs:
                raise ValueError(f'expected {self.num_envs} actions, got {actions.numel()}')
            if actions.dtype != torch.uint8 or actions.device != self.boards.device or not actions.is_contiguous():
                # (an int64 argmax, a host tensor ...: converted into a persistent uint8 buffer -- one copy kernel, no allocation)
                staged = self._actions_u8
                if staged is None:
                    staged = self._actions_u8 = torch.empty((self.num_envs,), dtype=torch.uint8, device=self.boards.device)
                staged.copy_(actions.reshape(self.num_envs))
                actions = staged
        dev, stream = _stream_handle(self.device)
        if self.step_counter is not None:  # the launch reads and advances the device counter: nothing to track on the host
            return self._step_with_counter(dev, stream, actions, flags, out_reward, out_done, out_legal, out_obs, out_max_exp,
                                           normalized_reward)
        if out_reward is None and out_done is None and out_legal is None and out_obs is None and out_max_exp is None \
                and not normalized_reward:  # the common, cheap path
            self.lib.step(dev, stream, self.num_envs, self.env_base, self._ptr_boards, _p(actions), self._ptr_reward,
                          self._ptr_done, self._ptr_legal, self._ptr_obs, self.seed, self._step_index, flags)
            self._step_index += 1
            return self.boards, self.reward, self.done
        if normalized_reward:
            if out_reward is None:
                raise ValueError('normalized_reward=True needs a float32 out_reward tensor (the persistent reward buffer is int32)')
            reward = self._check_out(out_reward, torch.float32, (self.num_envs,))
            flags |= capi.FLAG_REWARD_LOG2_F32
        else:
            reward = self.reward if out_reward is None else self._check_out(out_reward, torch.int32, (self.num_envs,))
        done = self.done if out_done is None else self._check_out(out_done, torch.uint8, (self.num_envs,))
        legal = self.legal if out_legal is None else self._check_out(out_legal, torch.uint8, (self.num_envs,))
        obs = self.obs if out_obs is None else self._check_out(out_obs, self.obs_dtype, (self.num_envs, 16))
        if out_max_exp is None and not normalized_reward:
            self.lib.step(dev, stream, self.num_envs, self.env_base, self._ptr_boards, _p(actions), _p(reward),
                          _p(done), _p(legal), _p(obs), self.seed, self._step_index, flags)
        else:  # the record form carries the extra recording output
            max_exp = None if out_max_exp is None else self._check_out(out_max_exp, torch.uint8, (self.num_envs,))
            record = (capi.StepArgs * 1)(capi.StepArgs(self.num_envs, self.env_base, self._ptr_boards, _p(actions), _p(reward), _p(done),
                                                       _p(legal), _p(obs), self.seed, self._step_index, flags, 0, _p(max_exp)))
            self.lib.step_many(dev, stream, record)
        self._step_index += 1
        return self.boards, reward, done

    def _step_with_counter(self, dev, stream, actions, flags, out_reward, out_done, out_legal, out_obs, out_max_exp, normalized_reward):
        capturing = self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if self._in_capture and not capturing and self._pending:
            pending, self._pending, self._in_capture = self._pending, 0, False
            raise RuntimeError(f'the captured region ended with {pending} step(s) not committed: call commit_steps() as the LAST '
                               'thing inside the capture, or every replay of the graph repeats the same step indices')
        if capturing and not self._in_capture and self._pending:
            raise RuntimeError('call commit_steps() before capturing a graph: the captured launches must start at offset 0 of '
                               'the device step counter, or every replay would skip steps')
        self._in_capture = capturing
        if normalized_reward:
            if out_reward is None:
                raise ValueError('normalized_reward=True needs a float32 out_reward tensor (the persistent reward buffer is int32)')
            reward = self._check_out(out_reward, torch.float32, (self.num_envs,))
            flags |= capi.FLAG_REWARD_LOG2_F32
        else:
            reward = self.reward if out_reward is None else self._check_out(out_reward, torch.int32, (self.num_envs,))
        done = self.done if out_done is None else self._check_out(out_done, torch.uint8, (self.num_envs,))
        legal = self.legal if out_legal is None else self._check_out(out_legal, torch.uint8, (self.num_envs,))
        obs = self.obs if out_obs is None else self._check_out(out_obs, self.obs_dtype, (self.num_envs, 16))
        max_exp = None if out_max_exp is None else self._check_out(out_max_exp, torch.uint8, (self.num_envs,))
        record = (capi.StepArgs * 1)(capi.StepArgs(self.num_envs, self.env_base, self._ptr_boards, _p(actions), _p(reward), _p(done),
                                                   _p(legal), _p(obs), self.seed, self._pending, flags, 0, _p(max_exp), _p(self.step_counter)))
        self.lib.step_many(dev, stream, record)
        self._pending += 1
        if not capturing:
            # eager step: advance the device counter NOW (one 1-thread launch), so that a graph replayed afterwards continues
            # behind this step instead of reusing its index; inside a capture the offsets accumulate until
            # commit_steps() closes the region
            self.commit_steps()
        return self.boards, reward, done

    def _record_step(self, actions_ptr, reward_ptr, done_ptr, legal_ptr, obs_ptr, max_exp_ptr, extra_flags: int) -> None:
        """The recorder's form of step(): the outputs are ROWS of a ``TrajectoryBuffer`` given as device addresses the buffer has
        checked once at construction (dtype, shape, device, contiguity), so a recorded move costs no tensor indexing and no
        validation on the host.  ``actions_ptr`` None = moves drawn in the kernel.  Not for envs with a device step counter."""
        flags = self.flags | extra_flags | (capi.FLAG_RANDOM_ACTIONS if actions_ptr is None else 0)
        dev, stream = _stream_handle(self.device)
        record = self._record_args
        if record is None:
            record = self._record_args = (capi.StepArgs * 1)()
        record[0] = capi.StepArgs(self.num_envs, self.env_base, self._ptr_boards, actions_ptr, reward_ptr, done_ptr, legal_ptr, obs_ptr,
                                  self.seed, self._step_index, flags, 0, max_exp_ptr)
        self.lib.step_many(dev, stream, record)
        self._step_index += 1

    def _check_out(self, t: torch.Tensor, dtype, shape) -> torch.Tensor:
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or t.device != self.boards.device or not t.is_contiguous():
            raise ValueError(f'output buffer must be a contiguous {dtype} tensor of shape {tuple(shape)} on {self.boards.device}, '
                             f'got {t.dtype} {tuple(t.shape)} on {t.device}')
        return t

    def random_actions(self, out: torch.Tensor, step_index: int | None = None) -> torch.Tensor:
        """Write into ``out`` (uint8, (N,)) the actions ``step(None)`` WOULD take at ``step_index`` (default: the next step):
        the low two bits of every environment's step word (env2048_random_actions).  ``step(out)`` then plays exactly the step
        ``step(None)`` would have played -- and the caller knows the moves (recorders need that)."""
        if self.step_counter is not None and step_index is None:
            raise RuntimeError('random_actions() needs an explicit step_index when the step counter lives on the device')
        out = self._check_out(out, torch.uint8, (self.num_envs,))
        dev, stream = _stream_handle(self.device)
        self.lib.random_actions(dev, stream, self.num_envs, self.env_base, self.seed,
                                self._step_index if step_index is None else int(step_index), _p(out))
        return out

    # launches up to this size are bound by the launch rate (3-4 us each), not by the kernel: K staged moves then go through ONE
    # env2048_replay launch (boards in registers, 6 B per board-step) -- measured crossover in profiles/r02_p_rollout_sizes.json
    REPLAY_MAX_ENVS = 1 << 18

    def step_many(self, actions: torch.Tensor, *, out_reward: torch.Tensor | None = None, out_done: torch.Tensor | None = None,
                  mode: str = 'auto'):
        """K staged moves: ``actions`` (K, N) uint8 -> the boards / rewards / done flags of K ``step(actions[k])`` calls.
        ``out_reward`` (K, N) int32 / ``out_done`` (K, N) uint8 receive the per-step rows (allocated when omitted).
        ``mode``: 'replay' = one env2048_replay launch (no per-step launch: the small-batch answer), 'launches' = K step
        launches enqueued by one env2048_step_sequence call, 'auto' = replay up to ``REPLAY_MAX_ENVS`` environments when only
        reward / done rows are wanted (the env was built without legal / obs outputs), launches otherwise.
        Returns ``(boards, reward_rows, done_rows)``; afterwards ``env.reward`` / ``env.done`` (and ``env.legal`` / ``env.obs``)
        hold the values of the LAST of the K moves, as after ``step``."""
        if actions.dtype != torch.uint8 or actions.dim() != 2 or actions.shape[1] != self.num_envs or not actions.is_contiguous() \
                or actions.device != self.boards.device:
            raise ValueError(f'actions must be a contiguous uint8 (K, {self.num_envs}) tensor on {self.boards.device}')
        if self.step_counter is not None:
            raise RuntimeError('step_many() keeps the step index on the host: build the env without device_step_counter')
        K, n = int(actions.shape[0]), self.num_envs
        reward = torch.empty((K, n), dtype=torch.int32, device=self.device) if out_reward is None else self._check_out(out_reward, torch.int32, (K, n))
        done = torch.empty((K, n), dtype=torch.uint8, device=self.device) if out_done is None else self._check_out(out_done, torch.uint8, (K, n))
        if mode not in ('auto', 'replay', 'launches'):
            raise ValueError("mode must be 'auto', 'replay' or 'launches'")
        plain = self.legal is None and self.obs is None
        use_replay = mode == 'replay' or (mode == 'auto' and plain and n <= self.REPLAY_MAX_ENVS)
        if use_replay and not plain:
            raise ValueError("mode='replay' writes reward / done rows only: this env also emits legal / obs per step")
        if K == 0:
            return self.boards, reward, done
        dev, stream = _stream_handle(self.device)
        if use_replay:
            self.lib.replay(dev, stream, n, self.env_base, self._ptr_boards, _p(actions), _p(reward), _p(done), n, self.seed,
                            self._step_index, K, capi.FLAG_AUTO_RESET if self.auto_reset else 0, None, None, None)
        else:
            import ctypes

            base = actions.data_ptr()
            table = (ctypes.c_void_p * K)(*[base + k * n for k in range(K)])
            self.lib.step_sequence(dev, stream, n, self.env_base, self._ptr_boards, table, _p(reward), _p(done), None, None, n,
                                   self.seed, self._step_index, K, self.flags)
            if self.legal is not None or self.obs is not None:  # the persistent legal / obs describe the LAST boards
                self.describe(self.legal, self.obs)
        # the persistent reward / done describe the LAST move, as after step() (they were left stale)
        self.reward.copy_(reward[K - 1])
        self.done.copy_(done[K - 1])
        self._step_index += K
        return self.boards, reward, done

    def rollout(self, num_steps: int):
        """``num_steps`` transitions with on-device random actions in ONE kernel (boards stay in
        registers).  Returns dict(return_sum int64, episodes uint32->int32 view, valid_moves)."""
        n = self.num_envs
        ret = torch.zeros((n,), dtype=torch.int64, device=self.device)
        eps = torch.zeros((n,), dtype=torch.int32, device=self.device)
        moves = torch.zeros((n,), dtype=torch.int32, device=self.device)
        dev, stream = _stream_handle(self.device)
        self.lib.rollout(dev, stream, n, self.env_base, _p(self.boards), self.seed, self.step_index, int(num_steps),
                         _p(ret), _p(eps), _p(moves), (capi.FLAG_AUTO_RESET if self.auto_reset else 0) | capi.FLAG_RANDOM_ACTIONS)
        self.step_index += int(num_steps)
        return {'return_sum': ret, 'episodes': eps, 'valid_moves': moves}

    def replay(self, actions: torch.Tensor, *, out_reward: torch.Tensor | None = None, out_done: torch.Tensor | None = None):
        """``actions`` (T, N) uint8 pre-staged moves played in ONE kernel (env2048_replay), boards held in registers:
        the same boards, rewards and done flags as T ``step(actions[t])`` calls.  ``out_reward`` (T, N) int32 /
        ``out_done`` (T, N) uint8 receive the per-step rows.  Returns dict(return_sum, episodes, valid_moves)."""
        if actions.dtype != torch.uint8 or actions.dim() != 2 or actions.shape[1] != self.num_envs or not actions.is_contiguous() \
                or actions.device != self.boards.device:
            raise ValueError(f'actions must be a contiguous uint8 (T, {self.num_envs}) tensor on {self.boards.device}')
        T, n = int(actions.shape[0]), self.num_envs
        if out_reward is not None:
            self._check_out(out_reward, torch.int32, (T, n))
        if out_done is not None:
            self._check_out(out_done, torch.uint8, (T, n))
        ret = torch.zeros((n,), dtype=torch.int64, device=self.device)
        eps = torch.zeros((n,), dtype=torch.int32, device=self.device)
        moves = torch.zeros((n,), dtype=torch.int32, device=self.device)
        dev, stream = _stream_handle(self.device)
        self.lib.replay(dev, stream, n, self.env_base, _p(self.boards), _p(actions), _p(out_reward), _p(out_done), n, self.seed,
                        self.step_index, T, capi.FLAG_AUTO_RESET if self.auto_reset else 0, _p(ret), _p(eps), _p(moves))
        self.step_index += T
        return {'return_sum': ret, 'episodes': eps, 'valid_moves': moves}

    def rollout_episodes(self, num_steps: int, stats: torch.Tensor | None = None) -> torch.Tensor:
        """``num_steps`` random-action transitions in one kernel, with per-EPISODE statistics accumulated on device
        (env2048_rollout_episodes): returns the 40-word uint64 block as an int64 tensor (see ``episode_statistics``).
        The game every environment is in is carried across CONSECUTIVE ``rollout_episodes`` calls, so games spanning calls are
        counted whole; ``reset()`` / ``set_boards()`` clear that carry (``reset_episode_carry``).  ``step`` / ``rollout`` /
        ``replay`` advance the boards without updating it: a game they end or begin in between is reported with the length
        and return accumulated by ``rollout_episodes`` calls only -- do not interleave them within one statistics run, or
        call ``reset_episode_carry()`` when you do.  Pass the tensor of a previous call as ``stats`` to keep accumulating."""
        n = self.num_envs
        if getattr(self, '_ep_length', None) is None:
            self._ep_length = torch.zeros((n,), dtype=torch.int32, device=self.device)
            self._ep_return = torch.zeros((n,), dtype=torch.int64, device=self.device)
        if stats is None:
            stats = torch.zeros((capi.ROLLOUT_STATS_WORDS,), dtype=torch.int64, device=self.device)
            stats[4] = -1  # = UINT64_MAX: the running minimum
        dev, stream = _stream_handle(self.device)
        self.lib.rollout_episodes(dev, stream, n, self.env_base, _p(self.boards), self.seed, self.step_index, int(num_steps),
                                  (capi.FLAG_AUTO_RESET if self.auto_reset else 0) | capi.FLAG_RANDOM_ACTIONS, _p(self._ep_length),
                                  _p(self._ep_return), _p(stats))
        self.step_index += int(num_steps)
        return stats

    @staticmethod
    def episode_statistics(stats) -> dict:
        """The report of evaluate_games (reinforce/training/self_play.py:503-521) from a statistics block."""
        w = [int(x) & 0xFFFFFFFFFFFFFFFF for x in torch.as_tensor(stats).cpu().tolist()]
        games = w[0]
        hist = {1 << e if e else 0: w[8 + e] for e in range(32) if w[8 + e]}
        if games == 0:
            return {'games': 0}
        mean_r, mean_l = w[2] / games, w[1] / games
        out = {
            'games': games,
            'mean_reward': mean_r,
            'std_reward': max(w[3] / games - mean_r * mean_r, 0.0) ** 0.5,
            'max_reward': float(w[5]),
            'min_reward': float(w[4]),
            'mean_max_tile': sum(t * c for t, c in hist.items()) / games,
            'max_tile': max(hist),
            'mean_length': mean_l,
            'std_length': max(w[6] / games - mean_l * mean_l, 0.0) ** 0.5,
            'max_tile_histogram': hist,
        }
        for tile in (2048, 4096, 8192, 16384, 32768):
            out[f'reached_{tile}'] = sum(c for t, c in hist.items() if t >= tile)
        return out

    def status(self):
        """done / legal bitmask / f32 obs / max exponent / empty count of the current boards."""
        n = self.num_envs
        out = {
            'done': torch.zeros((n,), dtype=torch.uint8, device=self.device),
            'legal': torch.zeros((n,), dtype=torch.uint8, device=self.device),
            'obs': torch.zeros((n, 16), dtype=torch.float32, device=self.device),
            'max_exp': torch.zeros((n,), dtype=torch.uint8, device=self.device),
            'n_empty': torch.zeros((n,), dtype=torch.uint8, device=self.device),
        }
        dev, stream = _stream_handle(self.device)
        self.lib.status(dev, stream, n, _p(self.boards), _p(out['done']), _p(out['legal']), _p(out['obs']),
                        _p(out['max_exp']), _p(out['n_empty']))
        return out

    # -- checkpoint / resume -----------------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        """Everything that determines the future of this batch -- the environment state is exactly the packed boards plus
        (seed, step index, first global index), and the game every environment is in when ``rollout_episodes`` statistics are
        being accumulated -- as host tensors / integers (``torch.save`` -able).  The reference never checkpoints its
        environment; a resumed run here continues the SAME random stream: stepping a restored batch gives
        byte for byte what the original would have produced."""
        carry = None
        if getattr(self, '_ep_length', None) is not None:
            carry = {'length': _host_copy(self._ep_length), 'return': _host_copy(self._ep_return)}
        return {'format': 1, 'num_envs': self.num_envs, 'seed': self.seed, 'env_base': self.env_base, 'step_index': int(self.step_index),
                'auto_reset': self.auto_reset, 'boards': _host_copy(self.boards), 'reward': _host_copy(self.reward), 'done': _host_copy(self.done),
                'episode_carry': carry}

    def load_state_dict(self, state: dict) -> None:
        """Restore ``state_dict()`` output into this batch (same number of environments; seed, index range and step index are
        taken from the checkpoint, so is the auto-reset policy)."""
        if state.get('format') != 1 or int(state['num_envs']) != self.num_envs:
            raise ValueError(f'checkpoint of {state.get("num_envs")} environments (format {state.get("format")}) does not fit this batch of {self.num_envs}')
        self.seed = int(state['seed']) & 0xFFFFFFFFFFFFFFFF
        self.env_base = int(state['env_base'])
        self.auto_reset = bool(state['auto_reset'])
        self.boards.copy_(torch.as_tensor(state['boards'], dtype=torch.uint8).reshape(self.num_envs, 16))
        self.reward.copy_(torch.as_tensor(state['reward'], dtype=torch.int32).reshape(self.num_envs))
        self.done.copy_(torch.as_tensor(state['done'], dtype=torch.uint8).reshape(self.num_envs))
        self.step_index = int(state['step_index'])
        self.reset_episode_carry()
        carry = state.get('episode_carry')
        if carry is not None:
            self._ep_length = torch.as_tensor(carry['length'], dtype=torch.int32).to(self.device).clone()
            self._ep_return = torch.as_tensor(carry['return'], dtype=torch.int64).to(self.device).clone()
        if self.legal is not None or self.obs is not None:
            self.describe(self.legal, self.obs)  # the persistent legal / obs describe the restored boards

    def set_boards(self, boards) -> None:
        """Overwrite the state with packed boards (anything convertible to an (N,16) uint8 tensor)."""
        b = torch.as_tensor(boards, dtype=torch.uint8).reshape(self.num_envs, 16)
        self.boards.copy_(b)
        self.reset_episode_carry()
