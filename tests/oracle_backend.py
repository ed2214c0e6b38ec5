"""
TEST-ONLY stand-in for ``simulate_2048_amd.capi.Library``: the same method signatures, but the
pointers are HOST pointers (CPU torch tensors / NumPy arrays) and the work is done by the CPU
oracle.  It exists so that the host-side logic of the package (buffer management, API mirrors,
sharding) can be exercised by ``-m "not gpu"`` tests in a container without a GPU.  The product
never imports it: ``simulate_2048_amd`` refuses CPU devices unless a library object is injected
explicitly, and only tests do that.
"""

from __future__ import annotations

import ctypes

import numpy as np

import oracle
from simulate_2048_amd import capi
from oracle import nxn as nxn_oracle


def _view(ptr, count, dtype):
    if ptr is None or count == 0:
        return None
    nbytes = count * np.dtype(dtype).itemsize
    buf = (ctypes.c_uint8 * nbytes).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype, count=count)


class OracleLibrary:
    """Drop-in for capi.Library over host memory. `device`/`stream` arguments are ignored."""

    def __init__(self):
        self.calls = []  # (name, n) log, handy for asserting launch counts in host-logic tests
        self.launches = []  # (name, device, stream, env_base, n) of the batch entry points: which device / stream a slice went to

    def abi_version(self):
        return 2

    def device_count(self):
        return 1

    def last_error(self):
        return ''

    def reset(self, device, stream, n, env_base, boards, done, seed, step_index):
        self.calls.append(('reset', n))
        self.launches.append(('reset', device, stream, env_base, n))
        b, _ = oracle.reset_batch(n, seed, step_index, env_base=env_base)
        _view(boards, n * 16, np.uint8)[:] = b.reshape(-1)
        if done is not None:
            _view(done, n, np.uint8)[:] = 0

    def step(self, device, stream, n, env_base, boards, actions, reward, done, legal, obs, seed, step_index, flags, max_exp=None):
        self.calls.append(('step', n))
        self.launches.append(('step', device, stream, env_base, n))
        b = _view(boards, n * 16, np.uint8)
        a = None if actions is None else _view(actions, n, np.uint8).copy()
        res = oracle.step_batch_ex(b.reshape(n, 16), a, seed, step_index, env_base=env_base, flags=flags & 3)
        b[:] = res['boards'].reshape(-1)
        if reward is not None:
            if flags & 64:  # REWARD_LOG2_F32
                _view(reward, n, np.float32)[:] = res['reward_norm']
            else:
                _view(reward, n, np.int32)[:] = res['reward']
        _view(done, n, np.uint8)[:] = res['done']
        if legal is not None:
            _view(legal, n, np.uint8)[:] = res['legal']
        if max_exp is not None:
            _view(max_exp, n, np.uint8)[:] = res['max_exp']
        if obs is not None:
            if flags & 4:  # OBS_F16
                _view(obs, n * 16, np.float16)[:] = res['obs'].reshape(-1).astype(np.float16)
            elif flags & 8:  # OBS_BF16: the upper half of the (exact) float32
                _view(obs, n * 16, np.uint16)[:] = (res['obs'].reshape(-1).view(np.uint32) >> 16).astype(np.uint16)
            else:
                _view(obs, n * 16, np.float32)[:] = res['obs'].reshape(-1)

    def step_many(self, device, stream, records):
        for r in records:
            step_index = r.step_index
            if r.step_counter:  # step index = device counter + offset
                step_index += int(_view(r.step_counter, 1, np.uint64)[0])
            self.step(device, stream, r.n, r.env_base, r.boards, r.actions, r.reward, r.done, r.legal, r.obs, r.seed, step_index,
                      r.flags & (15 | 64), max_exp=r.max_exp)

    def advance_counter(self, device, stream, counter, delta):
        _view(counter, 1, np.uint64)[0] += np.uint64(delta)

    def step_sequence(self, device, stream, n, env_base, boards, actions, reward, done, legal, obs, row_stride, seed, step_index,
                      count, flags):
        obs_bytes = 2 if flags & 12 else 4
        off = lambda p, t, scale: None if p is None else int(p) + t * row_stride * scale
        for t in range(count):
            self.step(device, stream, n, env_base, boards, None if actions is None else actions[t], off(reward, t, 4),
                      off(done, t, 1), off(legal, t, 1), off(obs, t, 16 * obs_bytes), seed, step_index + t, flags & (15 | 64))

    def step_state_i32(self, device, stream, n, env_base, boards, actions, step_count, done_io, total_reward, reward, max_tile,
                       seed, step_index):
        self.calls.append(('step_state_i32', n))
        zeros_i = np.zeros(n, dtype=np.int32)
        zeros_f = np.zeros(n, dtype=np.float32)
        res = oracle.step_state_i32(_view(boards, n * 16, np.int32).reshape(n, 16), _view(actions, n, np.int32),
                                    zeros_i if step_count is None else _view(step_count, n, np.int32), _view(done_io, n, np.uint8),
                                    zeros_f if total_reward is None else _view(total_reward, n, np.float32), seed, step_index,
                                    env_base=env_base)
        _view(boards, n * 16, np.int32)[:] = res['boards'].reshape(-1)
        _view(done_io, n, np.uint8)[:] = res['done']
        _view(reward, n, np.float32)[:] = res['reward']
        if step_count is not None:
            _view(step_count, n, np.int32)[:] = res['step_count']
        if total_reward is not None:
            _view(total_reward, n, np.float32)[:] = res['total_reward']
        if max_tile is not None:
            _view(max_tile, n, np.int32)[:] = res['max_tile']

    def step_state_i32_to(self, device, stream, n, env_base, boards_in, boards_out, actions, step_count_in, step_count_out,
                          done_in, done_out, total_in, total_out, reward, max_tile, seed, step_index):
        # copy in -> out, then the in-place form on the outputs (the inputs are never written)
        if boards_out != boards_in:
            _view(boards_out, n * 16, np.int32)[:] = _view(boards_in, n * 16, np.int32)
        if done_out != done_in:
            _view(done_out, n, np.uint8)[:] = _view(done_in, n, np.uint8)
        if step_count_out is not None and step_count_out != step_count_in:
            _view(step_count_out, n, np.int32)[:] = _view(step_count_in, n, np.int32)
        if total_out is not None and total_out != total_in:
            _view(total_out, n, np.float32)[:] = _view(total_in, n, np.float32)
        self.step_state_i32(device, stream, n, env_base, boards_out, actions, step_count_out, done_out, total_out, reward,
                            max_tile, seed, step_index)

    def step_state_i32_keyed(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                             done_in, done_out, total_in, total_out, reward, max_tile):
        self.calls.append(('step_state_i32_keyed', n))
        zeros_i = np.zeros(n, dtype=np.int32)
        zeros_f = np.zeros(n, dtype=np.float32)
        res = oracle.step_state_i32_keyed(_view(boards_in, n * 16, np.int32).reshape(n, 16), _view(actions, n, np.int32),
                                          zeros_i if step_count_in is None else _view(step_count_in, n, np.int32),
                                          _view(done_in, n, np.uint8), zeros_f if total_in is None else _view(total_in, n, np.float32),
                                          _view(keys, 2 * n, np.uint32))
        _view(boards_out, n * 16, np.int32)[:] = res['boards'].reshape(-1)
        _view(done_out, n, np.uint8)[:] = res['done']
        _view(reward, n, np.float32)[:] = res['reward']
        if step_count_out is not None:
            _view(step_count_out, n, np.int32)[:] = res['step_count']
        if total_out is not None:
            _view(total_out, n, np.float32)[:] = res['total_reward']
        if max_tile is not None:
            _view(max_tile, n, np.int32)[:] = res['max_tile']

    def reset_keyed_i32(self, device, stream, n, keys, boards):
        self.calls.append(('reset_keyed_i32', n))
        _view(boards, n * 16, np.int32)[:] = oracle.reset_keyed_i32(_view(keys, 2 * n, np.uint32)).reshape(-1)

    # the *_status forms = the plain call + the status of the boards it leaves behind (one launch on the GPU; here composed)
    def step_state_i32_keyed_status(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                    done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4):
        self.step_state_i32_keyed(device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out, done_in,
                                  done_out, total_in, total_out, reward, max_tile)
        self.calls[-1] = ('step_state_i32_keyed_status', n)
        if obs is not None or legal4 is not None:
            self.status_i32(device, stream, n, boards_out, obs, legal4, None)
            self.calls.pop()

    def reset_keyed_i32_status(self, device, stream, n, keys, boards, obs, legal4):
        self.reset_keyed_i32(device, stream, n, keys, boards)
        self.calls[-1] = ('reset_keyed_i32_status', n)
        if obs is not None or legal4 is not None:
            self.status_i32(device, stream, n, boards, obs, legal4, None)
            self.calls.pop()

    def step_state_i32_keyed_mirrored(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                      done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4, host_mirror_delta=0):
        assert host_mirror_delta == 0, 'the stand-in works on host memory: nothing to mirror'
        self.step_state_i32_keyed_status(device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                         done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4)

    def step_state_i32_keyed_inline(self, device, stream, n, host_keys, boards_in, boards_out, host_actions, step_count_in, step_count_out,
                                    done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4, host_mirror_delta=0,
                                    host_sequence=None, sequence_value=0):
        """Keys / actions come from host memory during the call (here everything is host memory): the _status form on them."""
        assert host_mirror_delta == 0 and host_sequence is None and n <= 64
        self.step_state_i32_keyed_status(device, stream, n, host_keys, boards_in, boards_out, host_actions, step_count_in, step_count_out,
                                         done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4)
        self.calls[-1] = ('step_state_i32_keyed_inline', n)

    def reset_state_i32_keyed(self, device, stream, n, keys, boards, step_count, done, total, obs, legal4, host_mirror_delta=0):
        assert host_mirror_delta == 0, 'the stand-in works on host memory: nothing to mirror'
        self.reset_keyed_i32_status(device, stream, n, keys, boards, obs, legal4)
        if step_count is not None:
            _view(step_count, n, np.int32)[:] = 0
        if done is not None:
            _view(done, n, np.uint8)[:] = 0
        if total is not None:
            _view(total, n, np.float32)[:] = 0.0

    def rollout(self, device, stream, n, env_base, boards, seed, step0, T, return_sum, episodes, valid_moves, flags):
        self.calls.append(('rollout', n))
        self.launches.append(('rollout', device, stream, env_base, n))
        b = _view(boards, n * 16, np.uint8)
        res = oracle.rollout_batch(b.reshape(n, 16), seed, step0, T, env_base=env_base, flags=flags)
        b[:] = res['boards'].reshape(-1)
        if return_sum is not None:
            _view(return_sum, n, np.int64)[:] = res['return_sum']
        if episodes is not None:
            _view(episodes, n, np.uint32)[:] = res['episodes']
        if valid_moves is not None:
            _view(valid_moves, n, np.uint32)[:] = res['valid_moves']

    def replay(self, device, stream, n, env_base, boards, actions, reward, done, row_stride, seed, step0, T, flags, return_sum, episodes,
               valid_moves):
        """By definition the sequence of single steps (include/env2048.h): composed from the oracle's step."""
        self.calls.append(('replay', n))
        b = _view(boards, n * 16, np.uint8)
        cur = b.reshape(n, 16).copy()
        acts = _view(actions, T * row_stride, np.uint8).reshape(T, row_stride)
        rew = None if reward is None else _view(reward, T * row_stride, np.int32).reshape(T, row_stride)
        dn = None if done is None else _view(done, T * row_stride, np.uint8).reshape(T, row_stride)
        ret, eps, moves = np.zeros(n, np.int64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        for t in range(T):
            before = cur
            res = oracle.step_batch(cur, acts[t, :n].copy(), seed, step0 + t, env_base=env_base, flags=flags & 1)
            cur = res['boards']
            ret += res['reward']
            eps += res['done']
            # a move was valid iff a tile was spawned: the afterstate differs from the board
            moves += oracle.afterstate_batch(before, acts[t, :n].copy())['changed']
            if rew is not None:
                rew[t, :n] = res['reward']
            if dn is not None:
                dn[t, :n] = res['done']
        b[:] = cur.reshape(-1)
        for ptr, val, dtype in ((return_sum, ret, np.int64), (episodes, eps, np.uint32), (valid_moves, moves, np.uint32)):
            if ptr is not None:
                _view(ptr, n, dtype)[:] = val

    def rollout_episodes(self, device, stream, n, env_base, boards, seed, step0, T, flags, ep_length, ep_return, stats):
        self.calls.append(('rollout_episodes', n))
        b = _view(boards, n * 16, np.uint8)
        st = _view(stats, oracle.ROLLOUT_STATS_WORDS, np.uint64)
        res = oracle.rollout_episodes(b.reshape(n, 16), seed, step0, T, env_base=env_base, flags=flags,
                                      ep_length=None if ep_length is None else _view(ep_length, n, np.uint32),
                                      ep_return=None if ep_return is None else _view(ep_return, n, np.int64), stats=st)
        b[:] = res['boards'].reshape(-1)
        st[:] = res['stats']
        if ep_length is not None:
            _view(ep_length, n, np.uint32)[:] = res['ep_length']
        if ep_return is not None:
            _view(ep_return, n, np.int64)[:] = res['ep_return']

    def _rows(self, ptr, ring_rows, row_stride, dtype):
        return None if ptr is None else _view(ptr, ring_rows * row_stride, dtype).reshape(ring_rows, row_stride)

    def count_episodes(self, device, stream, n, row_stride, rows, first_row, ring_rows, done, flags, counts):
        self.calls.append(('count_episodes', n))
        eps = oracle.segment_episodes(self._rows(done, ring_rows, row_stride, np.uint8), n=n, rows=rows, first_row=first_row,
                                      ring_rows=ring_rows, flags=flags)
        _view(counts, n, np.uint32)[:] = np.bincount(eps['env'], minlength=n).astype(np.uint32)

    def segment_episodes(self, device, stream, n, row_stride, rows, first_row, ring_rows, done, reward, max_exp, flags, offsets,
                         episodes, episode_of):
        self.calls.append(('segment_episodes', n))
        eps, eo = oracle.segment_episodes(self._rows(done, ring_rows, row_stride, np.uint8),
                                          self._rows(reward, ring_rows, row_stride, np.float32 if flags & 2 else np.int32),
                                          self._rows(max_exp, ring_rows, row_stride, np.uint8), n=n, rows=rows, first_row=first_row,
                                          ring_rows=ring_rows, flags=flags, want_episode_of=True)
        # the C-ABI writes environment i's records at offsets[i]: with the exclusive prefix sum of the counts that is the
        # oracle's own (environment, start) order
        off = _view(offsets, n, np.uint64)
        assert np.array_equal(off, np.concatenate([[0], np.cumsum(np.bincount(eps['env'], minlength=n))[:-1]]).astype(np.uint64))
        _view(episodes, len(eps) * 24, np.uint8)[:] = eps.view(np.uint8).reshape(-1)
        if episode_of is not None:
            out = self._rows(episode_of, ring_rows, row_stride, np.int32)
            touched = [(first_row + r) % ring_rows for r in range(rows)]
            out[touched, :n] = eo[touched, :n]

    def afterstate(self, device, stream, n, boards_in, actions, boards_out, reward, changed):
        self.calls.append(('afterstate', n))
        res = oracle.afterstate_batch(_view(boards_in, n * 16, np.uint8).reshape(n, 16).copy(), _view(actions, n, np.uint8).copy())
        _view(boards_out, n * 16, np.uint8)[:] = res['boards'].reshape(-1)
        if reward is not None:
            _view(reward, n, np.int32)[:] = res['reward']
        if changed is not None:
            _view(changed, n, np.uint8)[:] = res['changed']

    # -- boards of any size (env2048_nxn_*) ------------------------------------------------------------------------------------
    def nxn_afterstate(self, device, stream, n, size, boards_in, actions, boards_out, reward, changed):
        self.calls.append(('nxn_afterstate', n))
        cells = size * size
        res = nxn_oracle.afterstate_batch(_view(boards_in, n * cells, np.uint8).copy(), size, _view(actions, n, np.uint8).copy())
        _view(boards_out, n * cells, np.uint8)[:] = res['boards'].reshape(-1)
        if reward is not None:
            _view(reward, n, np.int32)[:] = res['reward']
        if changed is not None:
            _view(changed, n, np.uint8)[:] = res['changed']

    def nxn_step(self, device, stream, n, size, env_base, boards, actions, reward, done, legal, seed, step_index, flags):
        self.calls.append(('nxn_step', n))
        b = _view(boards, n * size * size, np.uint8)
        res = nxn_oracle.step_batch(b.copy(), size, None if actions is None else _view(actions, n, np.uint8).copy(), seed, step_index,
                                    env_base=env_base, flags=flags, want_legal=legal is not None)
        b[:] = res['boards'].reshape(-1)
        if reward is not None:
            _view(reward, n, np.int32)[:] = res['reward']
        _view(done, n, np.uint8)[:] = res['done']
        if legal is not None:
            _view(legal, n, np.uint8)[:] = res['legal']

    def nxn_status(self, device, stream, n, size, boards, done, legal, n_empty, max_exp):
        self.calls.append(('nxn_status', n))
        res = nxn_oracle.status_batch(_view(boards, n * size * size, np.uint8).copy(), size)
        for ptr, key in ((done, 'done'), (legal, 'legal'), (n_empty, 'n_empty'), (max_exp, 'max_exp')):
            if ptr is not None:
                _view(ptr, n, np.uint8)[:] = res[key]

    def nxn_reset(self, device, stream, n, size, env_base, boards, done, seed, step_index):
        self.calls.append(('nxn_reset', n))
        _view(boards, n * size * size, np.uint8)[:] = nxn_oracle.reset_batch(n, size, seed, step_index, env_base=env_base).reshape(-1)
        if done is not None:
            _view(done, n, np.uint8)[:] = 0

    def nxn_spawn_random(self, device, stream, n, size, env_base, boards, seed, step_index, count):
        self.calls.append(('nxn_spawn_random', n))
        b = _view(boards, n * size * size, np.uint8)
        b[:] = nxn_oracle.spawn_random_batch(b.copy(), size, seed, step_index, count, env_base=env_base).reshape(-1)

    def nxn_chance_outcomes(self, device, stream, n, size, boards, out_boards, out_count):
        self.calls.append(('nxn_chance_outcomes', n))
        cells = size * size
        out, count = nxn_oracle.chance_outcomes(_view(boards, n * cells, np.uint8).copy(), size)
        _view(out_boards, n * 2 * cells * cells, np.uint8)[:] = out.reshape(-1)
        _view(out_count, n, np.uint8)[:] = count

    def afterstates4(self, device, stream, n, boards, out_boards, out_reward, out_legal):
        self.calls.append(('afterstates4', n))
        b = _view(boards, n * 16, np.uint8).reshape(n, 16).copy()
        legal = np.zeros(n, dtype=np.uint8)
        ob = _view(out_boards, n * 64, np.uint8).reshape(n, 4, 16)
        for a in range(4):
            res = oracle.afterstate_batch(b, np.full(n, a, dtype=np.uint8))
            ob[:, a] = res['boards']
            if out_reward is not None:
                _view(out_reward, n * 4, np.int32).reshape(n, 4)[:, a] = res['reward']
            legal |= (res['changed'] << a).astype(np.uint8)
        if out_legal is not None:
            _view(out_legal, n, np.uint8)[:] = legal

    def spawn(self, device, stream, n, boards, spawn_k, spawn_e, placed):
        self.calls.append(('spawn', n))
        b = _view(boards, n * 16, np.uint8)
        res = oracle.spawn_batch(b.reshape(n, 16), _view(spawn_k, n, np.uint8).copy(), _view(spawn_e, n, np.uint8).copy())
        b[:] = res['boards'].reshape(-1)
        if placed is not None:
            _view(placed, n, np.uint8)[:] = res['placed']

    def spawn_random(self, device, stream, n, env_base, boards, seed, step_index, count):
        self.calls.append(('spawn_random', n))
        b = _view(boards, n * 16, np.uint8)
        b[:] = oracle.spawn_random_batch(b.reshape(n, 16), seed, step_index, count, env_base=env_base).reshape(-1)

    def status(self, device, stream, n, boards, done, legal, obs, max_exp, n_empty):
        self.calls.append(('status', n))
        res = oracle.status_batch(_view(boards, n * 16, np.uint8).reshape(n, 16).copy())
        for ptr, key, dtype, width in ((done, 'done', np.uint8, 1), (legal, 'legal', np.uint8, 1), (obs, 'obs', np.float32, 16),
                                       (max_exp, 'max_exp', np.uint8, 1), (n_empty, 'n_empty', np.uint8, 1)):
            if ptr is not None:
                _view(ptr, n * width, dtype)[:] = res[key].reshape(-1)

    def random_actions(self, device, stream, n, env_base, seed, step_index, actions):
        self.calls.append(('random_actions', n))
        out = _view(actions, n, np.uint8)
        for i in range(n):
            out[i] = oracle.step_word(seed, env_base + i, step_index) & 3

    def expectimax_depth2(self, device, stream, n, boards, terms, q, best_action, value):
        self.calls.append(('expectimax_depth2', n))
        res = oracle.expectimax_depth2(_view(boards, n * 16, np.uint8).reshape(n, 16).copy())
        if terms is not None:
            _view(terms, n * 16, np.int32)[:] = res['terms'].reshape(-1)
        if q is not None:
            _view(q, n * 4, np.float32)[:] = res['q'].astype(np.float32).reshape(-1)
        if best_action is not None:
            _view(best_action, n, np.uint8)[:] = res['best']
        if value is not None:
            _view(value, n, np.float32)[:] = res['value'].astype(np.float32)

    # ---- depth-d expectimax building blocks (restated with the oracle's afterstates; children in (action, cell, tile) order) ----
    def _afterstates(self, boards, n):
        b = _view(boards, n * 16, np.uint8).reshape(n, 16).copy()
        return b, [oracle.afterstate_batch(b, np.full(n, a, dtype=np.uint8)) for a in range(4)]

    def expand_count(self, device, stream, n, boards, counts):
        self.calls.append(('expand_count', n))
        _, after = self._afterstates(boards, n)
        c = np.zeros(n, dtype=np.uint32)
        for a in range(4):
            c += (2 * after[a]['changed'].astype(np.uint32) * (after[a]['boards'] == 0).sum(axis=1).astype(np.uint32))
        _view(counts, n, np.uint32)[:] = c

    def direct_stats(self, device):
        return {'available': False, 'reason': 'the oracle stand-in has no queues', 'calls': 0, 'steps': 0, 'packets': 0, 'declined': 0, 'reruns': 0,
                'last_chains': 0, 'last_gpu_ns': 0}

    def direct_eligible(self, n, count, flags, extras=False):
        return False

    def direct_plan(self, n, count, flags, extras=False):
        return {'eligible': False, 'chains': 0, 'parts': [], 'segments': 0, 'segment_steps': 0}

    def expand_scratch_words(self, n):
        return (n + capi.EXPAND_OFFSETS_BLOCK - 1) // capi.EXPAND_OFFSETS_BLOCK + 1

    def expand_offsets(self, device, stream, n, boards, offsets, scratch, host_total=None, sequence_value=0):
        self.calls.append(('expand_offsets', n))
        counts = np.zeros(n, dtype=np.uint32)
        self.expand_count(device, stream, n, boards, counts.ctypes.data)
        self.calls.pop()
        ends = np.cumsum(counts.astype(np.int64))
        _view(offsets, n, np.int64)[:] = ends - counts
        blocks = self.expand_scratch_words(n) - 1
        block_sums = np.add.reduceat(counts.astype(np.int64), np.arange(0, n, capi.EXPAND_OFFSETS_BLOCK))
        sc = _view(scratch, blocks + 1, np.int64)
        sc[:blocks] = np.cumsum(block_sums) - block_sums
        sc[blocks] = int(ends[-1])
        if host_total is not None:
            words = _view(host_total, 2, np.int64)
            words[0] = int(ends[-1])
            words[1] = sequence_value

    def wait_sequence(self, host_sequence, value, timeout_us):
        return int(_view(host_sequence, 1, np.uint64)[0]) == value

    def expand(self, device, stream, n, boards, offsets, children, reward, n_empty):
        self.calls.append(('expand', n))
        _, after = self._afterstates(boards, n)
        off = _view(offsets, n, np.int64)
        rw, ne = _view(reward, n * 4, np.int32).reshape(n, 4), _view(n_empty, n * 4, np.uint8).reshape(n, 4)
        total = 0
        for i in range(n):
            total += sum(2 * int((after[a]['boards'][i] == 0).sum()) for a in range(4) if after[a]['changed'][i])
        kids = _view(children, total * 16, np.uint8).reshape(total, 16) if total else None
        for i in range(n):
            at = int(off[i])
            for a in range(4):
                ok = bool(after[a]['changed'][i])
                cells = np.flatnonzero(after[a]['boards'][i] == 0) if ok else []
                rw[i, a] = int(after[a]['reward'][i]) if ok else 0
                ne[i, a] = len(cells)
                for cell in cells:  # row-major rank order
                    for e in (1, 2):
                        kids[at] = after[a]['boards'][i]
                        kids[at, cell] = e
                        at += 1

    def expectimax_depth2_f64(self, device, stream, n, boards, value):
        self.calls.append(('expectimax_depth2_f64', n))
        res = oracle.expectimax_depth2(_view(boards, n * 16, np.uint8).reshape(n, 16).copy())
        _view(value, n, np.float64)[:] = np.where(res['best'] == 255, 0.0, res['value'])

    def expectimax_backup(self, device, stream, n, reward, n_empty, offsets, child_value, value, q, best_action):
        self.calls.append(('expectimax_backup', n))
        rw, ne = _view(reward, n * 4, np.int32).reshape(n, 4), _view(n_empty, n * 4, np.uint8).reshape(n, 4)
        off = _view(offsets, n, np.int64)
        total = int(ne.astype(np.int64).sum()) * 2
        cv = _view(child_value, total, np.float64) if total else None
        val = _view(value, n, np.float64)
        for i in range(n):
            at, best_q, best_a = int(off[i]), 0.0, 255
            for a in range(4):
                qa = float('-inf')
                if ne[i, a]:
                    s2 = s4 = 0.0
                    for k in range(int(ne[i, a])):
                        s2 = s2 + float(cv[at])
                        s4 = s4 + float(cv[at + 1])
                        at += 2
                    qa = float(rw[i, a]) + (0.9 * s2 + 0.1 * s4) / float(ne[i, a])
                    if best_a == 255 or qa > best_q:
                        best_q, best_a = qa, a
                if q is not None:
                    _view(q, n * 4, np.float64)[4 * i + a] = qa
            val[i] = 0.0 if best_a == 255 else best_q
            if best_action is not None:
                _view(best_action, n, np.uint8)[i] = best_a

    def status_i32(self, device, stream, n, boards, obs, legal4, done):
        self.calls.append(('status_i32', n))
        res = oracle.status_batch(oracle.pack_values(_view(boards, n * 16, np.int32)).reshape(n, 16))
        if obs is not None:
            _view(obs, n * 16, np.float32)[:] = res['obs'].reshape(-1)
        if legal4 is not None:
            _view(legal4, n * 4, np.uint8)[:] = ((res['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8).reshape(-1)
        if done is not None:
            _view(done, n, np.uint8)[:] = (res['legal'] == 0).astype(np.uint8)  # JAX is_done: no legal action

    def afterstate_i32(self, device, stream, n, boards_in, actions, action, boards_out, reward, changed):
        self.calls.append(('afterstate_i32', n))
        packed = oracle.pack_values(_view(boards_in, n * 16, np.int32)).reshape(n, 16)
        acts = np.full(n, action & 3, dtype=np.uint8) if actions is None else (_view(actions, n, np.int32) & 3).astype(np.uint8)
        res = oracle.afterstate_batch(packed, acts)
        _view(boards_out, n * 16, np.int32)[:] = oracle.unpack_values(res['boards']).reshape(-1).astype(np.int32)
        if reward is not None:
            _view(reward, n, np.int32)[:] = res['reward']
        if changed is not None:
            _view(changed, n, np.uint8)[:] = res['changed']

    def chance_outcomes(self, device, stream, n, boards, out_boards, out_prob, out_count):
        self.calls.append(('chance_outcomes', n))
        res = oracle.chance_outcomes(_view(boards, n * 16, np.uint8).reshape(n, 16).copy())
        _view(out_boards, n * 32 * 16, np.uint8)[:] = res['boards'].reshape(-1)
        _view(out_prob, n * 32, np.float32)[:] = res['prob'].reshape(-1)
        if out_count is not None:
            _view(out_count, n, np.uint8)[:] = res['count']

    def expectimax_ply(self, device, stream, n, boards, succ_boards, succ_prob, succ_status, succ_count, after_reward, legal):
        self.calls.append(('expectimax_ply', n))
        res = oracle.expectimax_ply(_view(boards, n * 16, np.uint8).reshape(n, 16).copy())
        _view(succ_boards, n * 128 * 16, np.uint8)[:] = res['boards'].reshape(-1)
        _view(succ_prob, n * 128, np.float32)[:] = res['prob'].reshape(-1)
        for ptr, key, count, dtype in ((succ_status, 'status', 128, np.uint8), (succ_count, 'count', 4, np.uint8),
                                       (after_reward, 'reward', 4, np.int32), (legal, 'legal', 1, np.uint8)):
            if ptr is not None:
                _view(ptr, n * count, dtype)[:] = res[key].reshape(-1)

    def encode_onehot(self, device, stream, n, boards, width, out):
        self.calls.append(('encode_onehot', n))
        res = oracle.encode_onehot(_view(boards, n * 16, np.uint8).reshape(n, 16).copy(), width)
        _view(out, n * 16 * width, np.uint8)[:] = res.reshape(-1)

    def _pack(self, n, values, boards, dtype):
        _view(boards, n * 16, np.uint8)[:] = oracle.pack_values(_view(values, n * 16, dtype)).reshape(-1)

    def _unpack(self, n, boards, values, dtype):
        _view(values, n * 16, dtype)[:] = oracle.unpack_values(_view(boards, n * 16, np.uint8)).astype(dtype)

    def pack_i32(self, device, stream, n, values, boards):
        self._pack(n, values, boards, np.int32)

    def unpack_i32(self, device, stream, n, boards, values):
        self._unpack(n, boards, values, np.int32)

    def pack_i64(self, device, stream, n, values, boards):
        self._pack(n, values, boards, np.int64)

    def unpack_i64(self, device, stream, n, boards, values):
        self._unpack(n, boards, values, np.int64)
