"""
Boundary row b2: the reference's CALLERS run on the mirrors without edits.

The env-side statements of ``play_parallel_games`` (reinforce/training/self_play.py:298-300, 316-317, 346-356, 368,
374-375, 389 of the reference) and of the ``evaluate.py`` game loop (:89-128) are restated here in their own form --
``np.array(observations)``, ``np.array(dones)``, ``active_mask & ~dones_np``, ``bool(state.done)``,
``state._replace(...)``, ``float(reward)``, ``int(max_tile(...))`` -- with the MCTS policy replaced by a NumPy
random-legal choice and ``jax.random`` by a NumPy stand-in that hands out ``(N, 2)`` uint32 keys wrapped in an object
that offers ONLY ``__array__`` (what a JAX array on another backend looks like to us).  Every test runs on the
oracle-backed stand-in (host logic, CPU) and on the GPU box with the real HIP library, and is checked step by step
against the oracle's restatement of reinforce/game/env.py:79-133 under per-environment keys.
"""

import numpy as np
import pytest
import torch

import oracle
from simulate_2048_amd.interop import DeviceArray
from simulate_2048_amd.reinforce.game import core as game_core
from simulate_2048_amd.reinforce.game.core import encode_observation, is_done, legal_actions_mask, max_tile, next_state
from simulate_2048_amd.reinforce.game.env import (
    GameState,
    batched_get_legal_actions,
    batched_get_observation,
    batched_reset,
    batched_step,
    reset,
    step,
)


class ForeignArray:
    """An array from another framework: converts through ``__array__`` only (no torch, no DLPack)."""

    def __init__(self, a):
        self._a = np.asarray(a)

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __getitem__(self, i):
        return ForeignArray(self._a[i])

    @property
    def shape(self):
        return self._a.shape


class FakeJaxRandom:
    """Just enough of jax.random for the callers' key plumbing: split(key, n) -> (n, 2) uint32 keys."""

    @staticmethod
    def PRNGKey(seed):
        return ForeignArray(np.array([0, seed], dtype=np.uint32))

    @staticmethod
    def split(key, num=2):
        k = np.asarray(key).astype(np.uint64)
        rng = np.random.default_rng([int(k[0]), int(k[1]), 0x2048])
        return ForeignArray(rng.integers(0, 1 << 32, size=(num, 2), dtype=np.uint64).astype(np.uint32))


jrandom = FakeJaxRandom


def test_returns_are_numpy_convertible_device_arrays(api_runtime):
    keys = jrandom.split(jrandom.PRNGKey(1), 5)
    states = batched_reset(keys)
    for x in (states.board, states.done, states.step_count, states.total_reward, batched_get_observation(states),
              batched_get_legal_actions(states)):
        assert isinstance(x, DeviceArray) and isinstance(x, torch.Tensor)
        host = np.array(x)  # the caller's conversion; on the GPU this is the one device-to-host copy
        assert isinstance(host, np.ndarray) and host.shape == tuple(x.shape)
        assert isinstance(x[0], DeviceArray) and np.asarray(x[0]).shape == host.shape[1:]
    assert np.array(states.done).dtype == np.bool_ and np.array(states.board).dtype == np.int32
    # .dtype is the real torch.dtype: every torch consumer of a returned array works unchanged
    obs = batched_get_observation(states)
    for x, want in ((states.board, torch.int32), (states.done, torch.bool), (states.step_count, torch.int32),
                    (states.total_reward, torch.float32), (obs, torch.float32), (batched_get_legal_actions(states), torch.bool)):
        assert type(x.dtype) is torch.dtype and x.dtype is want and x.dtype in {want} and {want: 1}.get(x.dtype) == 1
        assert torch.zeros(1, dtype=x.dtype).dtype is want and torch.ones(2).to(x.dtype).dtype is want
        assert torch.promote_types(x.dtype, x.dtype) is want
    assert torch.finfo(obs.dtype).bits == 32 and torch.iinfo(states.board.dtype).bits == 32
    assert hasattr(states.board, '__dlpack__')
    back = torch.from_dlpack(states.board.as_subclass(torch.Tensor))
    assert back.data_ptr() == states.board.data_ptr()  # zero-copy hand-over of the device buffer
    # an integer scalar indexes / counts, a float one does not (as for any torch tensor: no silent truncation)
    assert list(range(states.step_count[0] + 2)) == [0, 1] and [7, 8][states.step_count[0]] == 7
    with pytest.raises(TypeError):
        range(states.total_reward[0])
    with pytest.raises(TypeError):
        [7, 8][states.total_reward[0]]


def test_batched_step_is_vmap_of_step(api_runtime):
    """reinforce/game/env.py:193-211: batched_step = vmap(step) -- row i depends on (states[i], actions[i], keys[i])
    only, and equals the single-environment call with that key (keys[1:] were ignored)."""
    n = 24
    keys = np.asarray(jrandom.split(jrandom.PRNGKey(7), n))
    states = batched_reset(ForeignArray(keys))
    for i in (0, 5, n - 1):
        single = reset(keys[i])
        assert np.array_equal(np.array(single.board), np.array(states.board[i]))
    rng = np.random.default_rng(3)
    for _ in range(6):
        actions = rng.integers(0, 4, size=n).astype(np.int32)
        step_keys = np.asarray(jrandom.split(ForeignArray(keys[0]), n))
        new_states, rewards, dones, info = batched_step(states, ForeignArray(actions), ForeignArray(step_keys))
        for i in (0, 3, 11, n - 1):
            one = GameState(states.board[i], states.step_count[i], states.done[i], states.total_reward[i])
            s1, r1, d1, info1 = step(one, int(actions[i]), step_keys[i])
            assert np.array_equal(np.array(s1.board), np.array(new_states.board[i]))
            assert float(r1) == float(rewards[i]) and bool(d1) == bool(dones[i])
            assert int(info1['max_tile']) == int(info['max_tile'][i]) and int(s1.step_count) == int(new_states.step_count[i])
        # other rows' keys do not leak: change every key but row 2's, row 2 is unchanged, and a permutation of the
        # batch permutes the result
        other = step_keys.copy()
        other[np.arange(n) != 2] ^= 0xA5A5
        alt, _, _, _ = batched_step(states, actions, other)
        assert np.array_equal(np.array(alt.board[2]), np.array(new_states.board[2]))
        perm = rng.permutation(n)
        sp = GameState(states.board[perm], states.step_count[perm], states.done[perm], states.total_reward[perm])
        pst, pr, _, _ = batched_step(sp, actions[perm], step_keys[perm])
        assert np.array_equal(np.array(pst.board), np.array(new_states.board)[perm])
        assert np.array_equal(np.array(pr), np.array(rewards)[perm])
        # and the whole batch equals the oracle's restatement under the same keys
        ref = oracle.step_state_i32_keyed(np.array(states.board).reshape(n, 16), actions, np.array(states.step_count),
                                          np.array(states.done), np.array(states.total_reward), step_keys)
        assert np.array_equal(np.array(new_states.board).reshape(n, 16), ref['boards'])
        assert np.array_equal(np.array(rewards), ref['reward']) and np.array_equal(np.array(dones), ref['done'].astype(bool))
        states, keys = new_states, step_keys
    with pytest.raises(ValueError):
        batched_step(states, actions, step_keys[:3])  # vmap would refuse mismatched leading axes too


def test_play_parallel_games_env_side_runs_unchanged(api_runtime):
    """The env-side statements of play_parallel_games, restated; the policy is a NumPy random-legal choice."""
    num_parallel, max_trajectory_length = 16, 600
    key = jrandom.PRNGKey(42)
    policy_rng = np.random.default_rng(0)

    # :298-300
    keys = jrandom.split(key, num_parallel + 1)
    key, game_keys = keys[0], keys[1:]
    states = batched_reset(game_keys)
    ref_boards = oracle.reset_keyed_i32(np.asarray(game_keys))
    assert np.array_equal(np.array(states.board).reshape(num_parallel, 16), ref_boards)
    ref_state = dict(step_count=np.zeros(num_parallel, np.int32), done=np.zeros(num_parallel, np.uint8),
                     total_reward=np.zeros(num_parallel, np.float32))

    all_observations = [[] for _ in range(num_parallel)]
    all_actions = [[] for _ in range(num_parallel)]
    all_rewards = [[] for _ in range(num_parallel)]
    active_mask = np.ones(num_parallel, dtype=bool)
    step_count = 0
    while np.any(active_mask) and step_count < max_trajectory_length:
        # :316-317
        observations = batched_get_observation(states)
        legal_masks = batched_get_legal_actions(states)
        # (MCTS + action selection replaced: uniformly random legal action, 0 where nothing is legal)
        legal_np = np.array(legal_masks)
        actions = np.array([policy_rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in legal_np], dtype=np.int32)
        # :346-349
        observations_np = np.array(observations)
        actions_np = np.array(actions)
        # :351-356
        key, step_key = jrandom.split(key)[0], jrandom.split(key)[1]
        step_keys = jrandom.split(step_key, num_parallel)
        states, rewards, dones, _ = batched_step(states, ForeignArray(actions), step_keys)
        rewards_np = np.array(rewards)
        dones_np = np.array(dones)
        # the same move through the oracle's restatement of env.py:79-133
        ref = oracle.step_state_i32_keyed(ref_boards, actions, ref_state['step_count'], ref_state['done'],
                                          ref_state['total_reward'], np.asarray(step_keys))
        ref_boards = ref['boards']
        ref_state = dict(step_count=ref['step_count'], done=ref['done'], total_reward=ref['total_reward'])
        assert np.array_equal(rewards_np, ref['reward']) and np.array_equal(dones_np, ref['done'].astype(bool))
        assert np.array_equal(observations_np.dtype, np.float32) and observations_np.shape == (num_parallel, 16)
        # :359-365
        for i in range(num_parallel):
            if active_mask[i]:
                all_observations[i].append(observations_np[i])
                all_actions[i].append(int(actions_np[i]))
                all_rewards[i].append(float(rewards_np[i]))
        # :368
        active_mask = active_mask & ~dones_np
        step_count += 1

    # :374-375, :389
    final_boards = np.array(states.board)
    final_dones = np.array(states.done)
    assert np.array_equal(final_boards.reshape(num_parallel, 16), ref_boards)
    assert final_dones.all() and not active_mask.any()  # random play ends well before 600 moves
    for i in range(num_parallel):
        assert len(all_observations[i]) == int(np.array(states.step_count)[i]) > 20
        assert float(sum(all_rewards[i])) == float(np.array(states.total_reward)[i])
        assert int(max_tile(final_boards[i])) == int(final_boards[i].max()) >= 16
        assert np.stack(all_observations[i], axis=0).shape == (len(all_actions[i]), 16)


def test_evaluate_game_loop_runs_unchanged(api_runtime):
    """The game loop of reinforce/evaluate.py:89-128, restated; two games."""
    key = jrandom.PRNGKey(5)
    policy_rng = np.random.default_rng(1)
    max_tiles, total_rewards, episode_lengths = [], [], []
    for game_num in range(2):
        key, game_key = jrandom.split(key)[0], jrandom.split(key)[1]
        state = reset(game_key)
        cumulative_reward = 0.0
        steps = 0
        while not bool(state.done):
            k4 = jrandom.split(key, 4)
            key, mcts_key, action_key, step_key = k4[0], k4[1], k4[2], k4[3]
            obs = encode_observation(state.board)
            legal_mask = legal_actions_mask(state.board)
            assert np.array(obs).shape == (16,) and np.array(legal_mask).dtype == np.bool_
            action = int(policy_rng.choice(np.flatnonzero(np.array(legal_mask))))
            before = np.array(state.board).copy()
            new_board, reward = next_state(state.board, action, step_key)
            done = is_done(new_board)
            cumulative_reward += float(reward)
            steps += 1
            # the transition against the oracle: afterstate of the chosen (legal) action + exactly one new 2/4 tile
            after, score = oracle.latent_state(oracle.pack_values(before).reshape(16), action)
            diff = np.flatnonzero(oracle.pack_values(np.array(new_board)).reshape(16) != after)
            assert len(diff) == 1 and after[diff[0]] == 0 and int(np.array(new_board).reshape(16)[diff[0]]) in (2, 4)
            assert float(reward) == float(score)
            state = state._replace(board=new_board, done=done)
            current_max = int(max_tile(state.board))
            assert current_max == int(np.array(state.board).max())
        max_tiles.append(int(max_tile(state.board)))
        total_rewards.append(cumulative_reward)
        episode_lengths.append(steps)
        assert not np.array(game_core.legal_actions_mask(state.board)).any()
    assert all(t >= 16 for t in max_tiles) and all(n > 20 for n in episode_lengths) and all(r > 0 for r in total_rewards)


# ---------------------------------------------------------------------------------------------------------------------------
# One launch per move, allocation-free (VERDICT r2 "next" #4): result sets are reused only when nobody can see them any more,
# the observation / legal mask of a new state come out of the launch that produced it, small host inputs go through pinned memory.
# ---------------------------------------------------------------------------------------------------------------------------
from simulate_2048_amd import resultpool  # noqa: E402


def _loop_keys(rng, n):
    return rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)


def test_one_launch_per_move_and_no_new_result_sets(api_runtime):
    """The restated inner loop of play_parallel_games (self_play.py:316-317,354): after warm-up a move costs ONE C-ABI call
    (no separate status launches) and creates no result set; results equal the three separate calls of round 2."""
    n = 8
    rng = np.random.default_rng(3)
    resultpool.pool.clear()
    states = batched_reset(_loop_keys(rng, n))
    calls = getattr(api_runtime.lib, 'calls', None)
    for move in range(40):
        observations = batched_get_observation(states)
        legal_masks = batched_get_legal_actions(states)
        # the same answers from a launch of their own on a COPY of the board (not a pooled tensor: takes the general path)
        plain = GameState(torch.as_tensor(np.array(states.board)), states.step_count, states.done, states.total_reward)
        assert np.array_equal(np.array(observations), np.array(batched_get_observation(plain)))
        assert np.array_equal(np.array(legal_masks), np.array(batched_get_legal_actions(plain)))
        legal_np = np.array(legal_masks)
        actions = np.array([np.flatnonzero(m)[0] if m.any() else 0 for m in legal_np], dtype=np.int32)
        if move == 10:
            created = resultpool.pool.created
            mark = len(calls) if calls is not None else 0
        states, rewards, dones, info = batched_step(states, actions, _loop_keys(rng, n))
        assert np.array_equal(np.array(info['step_count']), np.array(states.step_count))
    assert resultpool.pool.created == created, 'the loop must reuse its result sets once it has two or three'
    if calls is not None:  # host-logic run: count what reached the library from move 10 on
        names = [c[0] for c in calls[mark:]]
        assert names.count('step_state_i32_keyed_status') == 30
        # (status_i32 appears only for the `plain` cross-check above: two per move, moves 11 .. 39)
        assert names.count('status_i32') == 58 and len(names) == 88


def test_held_results_and_views_are_never_overwritten(api_runtime):
    """A state (or just a VIEW of one of its arrays) that the caller still holds keeps its values while the loop goes on."""
    n = 4
    rng = np.random.default_rng(9)
    resultpool.pool.clear()
    states = batched_reset(_loop_keys(rng, n))
    kept_state = states
    kept_board = np.array(states.board).copy()
    kept_obs_view = batched_get_observation(states)[1]  # a view into the set's memory, the parent object is dropped
    kept_obs = np.array(kept_obs_view).copy()
    history = []
    for move in range(25):
        states, rewards, dones, _ = batched_step(states, np.full(n, move & 3, dtype=np.int32), _loop_keys(rng, n))
        history.append((states, np.array(states.board).copy(), rewards, np.array(rewards).copy()))  # keep EVERYTHING alive
    assert np.array_equal(np.array(kept_state.board), kept_board) and np.array_equal(np.array(kept_obs_view), kept_obs)
    for st, board, rw, rw_np in history:
        assert np.array_equal(np.array(st.board), board) and np.array_equal(np.array(rw), rw_np)
    assert len({id(st.board) for st, *_ in history}) == len(history)  # 25 live states -> 25 different arrays
    del history, kept_state, kept_obs_view
    # everything released: the pool serves the loop again without growing
    created = resultpool.pool.created
    for move in range(10):
        states, *_ = batched_step(states, np.zeros(n, dtype=np.int32), _loop_keys(rng, n))
    assert resultpool.pool.created <= created + 1


def test_stash_is_dropped_when_the_board_is_edited_in_place(api_runtime):
    n = 3
    rng = np.random.default_rng(4)
    resultpool.pool.clear()
    states = batched_reset(_loop_keys(rng, n))
    before = np.array(batched_get_observation(states))
    states.board[0, 0, 0] = 64  # a caller poking the board (the reference's tests do: test_environment.py:106-107)
    after = np.array(batched_get_observation(states))
    assert after[0, 0] == 6 / 16 and np.array_equal(after.reshape(-1)[1:], np.array(before).reshape(-1)[1:]) or before[0, 0] == 6 / 16
    legal = np.array(batched_get_legal_actions(states))
    ref = oracle.status_batch(oracle.pack_values(np.array(states.board).reshape(n, 16)).reshape(n, 16))
    assert np.array_equal(legal, ((ref['legal'][:, None] >> np.arange(4)) & 1).astype(bool))


def test_host_stager_ring_wraps_without_losing_data():
    st = resultpool.HostStager(torch.device('cpu'), capacity=256)
    seen = []
    for i in range(40):
        a = np.arange(i, i + 5, dtype=np.uint32)
        ptr = st.stage(a)
        assert ptr % 16 == 0 and st.base <= ptr and ptr + a.nbytes <= st.base + st.capacity
        off = ptr - st.base
        assert np.array_equal(st.host[off:off + a.nbytes].view(np.uint32), a)
        seen.append(off)
    assert min(seen) == 0 and max(seen) >= 128  # both halves were used


def test_result_pool_forgets_old_shapes(api_runtime):
    """A caller that cycles through many batch sizes does not pin a result set per size for ever: beyond MAX_KEYS shapes the
    least recently used shape's sets (and their registry entries) are dropped; states the caller still holds stay valid."""
    rng = np.random.default_rng(11)
    resultpool.pool.clear()
    kept = batched_reset(_loop_keys(rng, 3))  # shape n = 3, held by the caller for the whole test
    kept_board = np.array(kept.board).copy()
    for n in range(4, 4 + resultpool.MAX_KEYS + 5):
        st = batched_reset(_loop_keys(rng, n))
        st, *_ = batched_step(st, np.zeros(n, dtype=np.int32), _loop_keys(rng, n))
    assert len(resultpool.pool._sets) <= resultpool.MAX_KEYS
    assert len(resultpool.pool.registry) <= resultpool.MAX_KEYS * resultpool.MAX_SETS_PER_KEY * 8
    assert np.array_equal(np.array(kept.board), kept_board)  # dropped from the pool, never overwritten
    obs = batched_get_observation(kept)  # no longer a pooled state: answered by a status launch of its own
    assert np.array_equal(np.array(obs), oracle.status_batch(oracle.pack_values(kept_board.reshape(3, 16)).reshape(3, 16))['obs'])
    again, *_ = batched_step(kept, np.zeros(3, dtype=np.int32), _loop_keys(rng, 3))
    assert np.array(again.board).shape == (3, 4, 4)


def test_stash_comes_back_when_an_edited_set_is_reused(api_runtime):
    """An in-place edit drops the stash of THAT state only: once the set is handed out again its launch re-validates it
    (a set whose stash stayed invalid for ever would cost two status launches per move from then on)."""
    n = 4
    rng = np.random.default_rng(21)
    resultpool.pool.clear()
    states = batched_reset(_loop_keys(rng, n))
    states.step_count.add_(0)  # a caller's own bookkeeping on a field: the whole set shares one version counter
    assert resultpool.pool.lookup(states.board)[0].stash_valid() is False
    calls = getattr(api_runtime.lib, 'calls', None)
    for move in range(12):
        states, *_ = batched_step(states, np.full(n, move & 3, dtype=np.int32), _loop_keys(rng, n))
    mark = len(calls) if calls is not None else 0
    for move in range(12):  # every set of the pool has been rewritten by now, the edited one included
        assert resultpool.pool.lookup(states.board)[0].stash_valid()
        batched_get_observation(states), batched_get_legal_actions(states)
        states, *_ = batched_step(states, np.full(n, move & 3, dtype=np.int32), _loop_keys(rng, n))
    if calls is not None:
        assert [c[0] for c in calls[mark:]].count('status_i32') == 0


def test_acquire_leases_a_set_to_one_thread_at_a_time():
    """Two threads calling the mirrors concurrently never get the same idle set: a set is leased from acquire() until the
    mirror has built the state it returns; and sets are kept per stream (reuse is only ordered on the stream of the last use)."""
    p = resultpool.ResultPool()
    dev = torch.device('cpu')
    a = p.acquire(5, dev, 0)
    b = p.acquire(5, dev, 0)  # `a` is idle by reference counts (nobody holds its tensors yet) but leased
    assert a is not b and a.leased and b.leased and not a.idle()
    a.leased = False
    assert p.acquire(5, dev, 0) is a  # released without anybody holding it: free again
    a.leased = False
    other_stream = p.acquire(5, dev, 1234)
    assert other_stream is not a and other_stream is not b
    import threading

    got, barrier = [], threading.Barrier(4)

    def worker():
        barrier.wait()
        for _ in range(200):
            rs = p.acquire(7, dev, 0)
            got.append(id(rs))
            held = rs.board  # what the mirror does: take a reference, then end the lease
            assert rs.leased
            rs.leased = False
            del held

    threads = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert len(got) == 800


def test_mirror_loops_from_several_threads(api_runtime):
    """Three self-play loops at the reference's batch sizes run concurrently (each on its own stream on the GPU), sharing the
    process-wide result pool: every move of every loop equals the oracle's step under the same keys -- no loop ever sees a
    result set another loop was handed."""
    import contextlib
    import threading

    resultpool.pool.clear()
    errors = []

    def loop(tid, n):
        try:
            rng = np.random.default_rng(100 + tid)
            stream = torch.cuda.Stream(api_runtime.device) if api_runtime.device.type == 'cuda' else None
            ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
            with ctx:
                states = batched_reset(_loop_keys(rng, n))
                for move in range(60):
                    legal = np.array(batched_get_legal_actions(states))
                    actions = np.array([np.flatnonzero(m)[0] if m.any() else 0 for m in legal], dtype=np.int32)
                    keys = _loop_keys(rng, n)
                    before = (np.array(states.board).reshape(n, 16), np.array(states.step_count), np.array(states.done).astype(np.uint8),
                              np.array(states.total_reward))
                    states, rewards, dones, _ = batched_step(states, actions, keys)
                    want = oracle.step_state_i32_keyed(before[0], actions, before[1], before[2], before[3], np.asarray(keys).astype(np.uint32))
                    assert np.array_equal(np.array(states.board).reshape(n, 16), want['boards']), (tid, move)
                    assert np.array_equal(np.array(rewards), want['reward']) and np.array_equal(np.array(dones), want['done'].astype(bool)), (tid, move)
        except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
            errors.append(exc)

    threads = [threading.Thread(target=loop, args=(i, n)) for i, n in enumerate((8, 8, 4))]  # two loops share a shape
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[0]


@pytest.mark.gpu
def test_numpy_reads_of_a_state_come_from_the_host_mirror():
    """The np.array(...) conversions the reference's loop makes per move (self_play.py:346-356) cost NO device-to-host copy:
    the launch wrote the set's pinned host mirror itself.  Values equal the plain per-array copies; what the caller gets is
    its own copy; an in-place torch edit is seen by the next read (one refresh copy); views take the general path."""
    n = 8
    rng = np.random.default_rng(31)
    resultpool.pool.clear()
    states = batched_reset(_loop_keys(rng, n))
    rs0 = resultpool.pool.lookup(states.board)[0]
    assert rs0.mirror_delta != 0 and rs0.mirror_delta % 16 == 0
    fresh = [np.array(states.step_count), np.array(states.done), np.array(states.total_reward), np.array(states.board)]
    assert not fresh[0].any() and not fresh[1].any() and not fresh[2].any() and (fresh[3] != 0).sum() == 2 * n and rs0.mirror_copies == 0
    for move in range(12):
        states, rewards, dones, info = batched_step(states, np.full(n, move & 3, dtype=np.int32), _loop_keys(rng, n))
        rs = resultpool.pool.lookup(states.board)[0]
        before = rs.mirror_copies
        fields = (rewards, dones, batched_get_observation(states), batched_get_legal_actions(states), states.board, info['max_tile'],
                  states.step_count, states.total_reward)
        got = [np.array(t) for t in fields]
        assert rs.mirror_copies == before, 'reads of a freshly written state need no copy'
        plain = [t.as_subclass(torch.Tensor).cpu().numpy() for t in fields]
        for a, b in zip(got, plain):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        got[0][:] = -1.0  # the caller's array is its own
        assert np.array_equal(np.array(rewards), plain[0])
    states.board[0, 0, 0] = 64  # an in-place edit is seen by the next read (one refresh copy of the set)
    assert np.array(states.board)[0, 0, 0] == 64 and rs.mirror_copies == before + 1
    assert np.array(states.board)[0, 0, 0] == 64 and rs.mirror_copies == before + 1
    view = states.board[1]  # a view is not a pooled tensor: the general path, same values
    assert np.array_equal(np.array(view), np.array(states.board)[1])
    assert np.array(states.board, dtype=np.int64).dtype == np.int64
    # a large batch is not mirrored (its arrays are read one by one as before)
    big = batched_reset(_loop_keys(rng, 4096))
    assert resultpool.pool.lookup(big.board)[0].mirror_delta == 0 and np.array(big.board).shape == (4096, 4, 4)


@pytest.mark.gpu
def test_the_fast_move_path_is_present_on_this_box():
    """the 18-microsecond move depends on a private torch symbol (``torch._C._storage_Use_Count``: without it
    no result set is ever reused), on pinned-memory mirrors and on the completion word.  None of them failing is an error --
    the mirrors fall back to slower paths -- so THIS test fails loudly when a torch upgrade (or anything else) takes the fast
    path away.  After a warm-up, one move of the reference's loop at N = 8 (self_play.py:316-317,346-356: two getters, the
    step, four np.array read-backs): no new ResultSet, exactly ONE C-ABI launch, no device-to-host copy, no stream
    synchronisation (the read-backs poll the launch's completion word)."""
    from simulate_2048_amd import runtime

    assert resultpool._use_count is not None, 'torch._C._storage_Use_Count is gone: result sets are never reused (3x slower moves)'
    n = 8
    rng = np.random.default_rng(77)
    resultpool.pool.clear()
    rt = runtime.current()
    calls = []

    class Spy:  # counts the C-ABI launches the mirrors make (wait_sequence is a host-side poll, not a launch)
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if not callable(fn) or name == 'wait_sequence':
                return fn

            def counted(*args, **kwargs):
                calls.append(name)
                return fn(*args, **kwargs)

            return counted

    with runtime.use(Spy(rt.lib), rt.device):
        states = batched_reset(_loop_keys(rng, n))

        def one_move(states):
            observations = batched_get_observation(states)
            legal_masks = batched_get_legal_actions(states)
            host_obs, host_legal = np.array(observations), np.array(legal_masks)
            del observations, legal_masks
            new_states, rewards, dones, _ = batched_step(states, (host_legal.argmax(axis=1)).astype(np.int32), _loop_keys(rng, n))
            return new_states, host_obs, np.array(rewards), np.array(dones)

        for _ in range(10):  # warm-up: the pool reaches its steady state (two sets alternate)
            states, *_ = one_move(states)
        created = resultpool.pool.created
        sets = {id(rs): rs for sets in resultpool.pool._sets.values() for rs in sets}
        before = {k: (rs.mirror_copies, rs.stream_syncs, rs.signal_hits) for k, rs in sets.items()}
        calls.clear()
        moves = 50
        for _ in range(moves):
            prev = np.array(states.board).reshape(n, 16)
            states, host_obs, host_rewards, host_dones = one_move(states)
        assert resultpool.pool.created == created, 'a steady-state move allocated a new ResultSet'
        assert calls == ['step_state_i32_keyed_inline'] * moves, calls[:6]  # (host-side keys / actions: the inline form)
        copies = sum(rs.mirror_copies - before[k][0] for k, rs in sets.items())
        syncs = sum(rs.stream_syncs - before[k][1] for k, rs in sets.items())
        hits = sum(rs.signal_hits - before[k][2] for k, rs in sets.items())
        assert copies == 0, f'{copies} device-to-host copies for read-backs the launch should have mirrored'
        assert hits + syncs == moves, (hits, syncs)  # one wait per launch: the first read-back of its results ...
        assert syncs <= moves // 10, f'{syncs} of {moves} moves fell back to synchronising the stream'  # ... found the completion word (a rare timeout is legal)
        assert all(rs.seq_ptr is not None and rs.mirror_delta != 0 for rs in sets.values())
    torch.cuda.synchronize()


def test_evaluate_games_loop_on_the_core_functions(api_runtime):
    """The env-side statements of evaluate_games (reinforce/evaluate.py:88-128) and play_game (reinforce/training/self_play.py:
    95-135), restated: per move encode_observation(state.board), legal_actions_mask(state.board), next_state(state.board,
    action, step_key), is_done(new_board), float(reward), state._replace(board=new_board, done=done); max_tile(state.board) at
    the end.  On the mirrors next_state is ONE launch that also leaves observation / legal mask / terminal flag / max tile of
    the new board with it, so the other calls launch nothing; every value equals the oracle's keyed step."""
    resultpool.pool.clear()
    calls = getattr(api_runtime.lib, 'calls', None)
    for game in range(3):
        state = reset(np.array([game, 99], dtype=np.uint32))
        ref = np.array(state.board).reshape(1, 16).astype(np.int32)
        cumulative, steps = 0.0, 0
        mark = len(calls) if calls is not None else 0
        while not bool(state.done) and steps < 60:
            obs = encode_observation(state.board)
            legal_mask = legal_actions_mask(state.board)
            status = oracle.status_batch(oracle.pack_values(ref).reshape(1, 16))
            assert np.array_equal(np.array(obs), status['obs'][0]) and np.array(obs).dtype == np.float32
            legal = np.array(legal_mask)
            assert legal.dtype == np.bool_ and [int(x) for x in legal] == [(int(status['legal'][0]) >> a) & 1 for a in range(4)]
            action = int(np.flatnonzero(legal)[-1]) if legal.any() else 0
            step_key = np.array([steps, 7 + game], dtype=np.uint32)
            new_board, reward = next_state(state.board, action, step_key)
            done = is_done(new_board)
            want = oracle.step_state_i32_keyed(ref, np.array([action], np.int32), np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(1, np.float32),
                                               step_key.reshape(1, 2))
            ref = want['boards']
            assert np.array_equal(np.array(new_board).reshape(1, 16), ref) and float(reward) == float(want['reward'][0])
            assert bool(done) == bool(want['done'][0]) and int(max_tile(new_board)) == int(want['max_tile'][0])
            cumulative += float(reward)
            steps += 1
            state = state._replace(board=new_board, done=done)
        assert steps > 10 and int(max_tile(state.board)) == int(ref.max())
        if calls is not None:  # host logic: from the first move on, ONE library call per move
            names = [c[0] for c in calls[mark:]]
            assert names == ['step_state_i32_keyed_inline'] * steps, names[:6]
    # boards that are nobody's alias take the general path: the same answers
    plain = torch.as_tensor(ref.reshape(4, 4))
    status = oracle.status_batch(oracle.pack_values(ref).reshape(1, 16))
    assert np.array_equal(np.array(encode_observation(plain)), status['obs'][0])
    assert [int(x) for x in np.array(legal_actions_mask(plain))] == [(int(status['legal'][0]) >> a) & 1 for a in range(4)]
    assert bool(is_done(plain)) == (int(status['legal'][0]) == 0) and int(max_tile(plain)) == int(ref.max())
    b2, r2 = next_state(plain, 1, 12345)
    want = oracle.step_state_i32_keyed(ref, np.array([1], np.int32), np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(1, np.float32),
                                       np.array([[0, 12345]], dtype=np.uint32))
    assert np.array_equal(np.array(b2).reshape(1, 16), want['boards']) and float(r2) == float(want['reward'][0])


@pytest.mark.gpu
def test_single_game_loop_runs_on_the_one_environment_result_set():
    """The reference's SINGLE-game loops (reinforce/training/self_play.py:84-135, reinforce/evaluate.py:89-128): per move
    get_observation, get_legal_actions, step with a fresh key, bool(done), float(reward).  States that come out of reset /
    step are the un-batched aliases of a pooled one-environment result set, so a move is ONE launch (host key and action in the
    kernel arguments), the getters return the stash, and every read-back -- np.array, bool, float, int -- comes from the host
    mirror: no device-to-host copy, no new result set, results equal to the oracle's keyed step."""
    from simulate_2048_amd import runtime
    from simulate_2048_amd.reinforce.game.env import get_legal_actions, get_observation

    resultpool.pool.clear()
    rt = runtime.current()
    calls = []

    class Spy:
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if not callable(fn) or name == 'wait_sequence':
                return fn

            def counted(*args, **kwargs):
                calls.append(name)
                return fn(*args, **kwargs)

            return counted

    with runtime.use(Spy(rt.lib), rt.device):
        state = reset(np.array([5, 77], dtype=np.uint32))
        assert tuple(state.board.shape) == (4, 4) and state.board.dtype == torch.int32 and tuple(state.done.shape) == ()
        board = np.array(state.board)
        assert (board != 0).sum() == 2 and int(state.step_count) == 0 and bool(state.done) is False and float(state.total_reward) == 0.0
        ref_board = board.reshape(1, 16).astype(np.int32)
        ref_steps, ref_done, ref_total = np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(1, np.float32)
        for move in range(40):
            if move == 10:
                created, copies_before = resultpool.pool.created, sum(rs.mirror_copies for sets in resultpool.pool._sets.values() for rs in sets)
                calls.clear()
            obs = np.array(get_observation(state))
            legal = np.array(get_legal_actions(state))
            assert obs.shape == (16,) and obs.dtype == np.float32 and legal.shape == (4,) and legal.dtype == np.bool_
            action = int(np.flatnonzero(legal)[0]) if legal.any() else 0
            key = np.array([move, 4242], dtype=np.uint32)
            state, reward, done, info = step(state, action, key)
            want = oracle.step_state_i32_keyed(ref_board, np.array([action], np.int32), ref_steps, ref_done, ref_total, key.reshape(1, 2))
            ref_board, ref_steps, ref_done, ref_total = want['boards'], want['step_count'], want['done'], want['total_reward']
            assert np.array_equal(np.array(state.board).reshape(1, 16), ref_board) and float(reward) == float(want['reward'][0])
            assert bool(done) == bool(want['done'][0]) and int(info['max_tile']) == int(want['max_tile'][0]) and int(state.step_count) == int(ref_steps[0])
            status = oracle.status_batch(oracle.pack_values(ref_board).reshape(1, 16))
            assert np.array_equal(np.array(get_observation(state)), status['obs'][0])
        assert resultpool.pool.created == created, 'a steady-state single-game move allocated a result set'
        assert calls == ['step_state_i32_keyed_inline'] * 30, calls[:5]
        assert sum(rs.mirror_copies for sets in resultpool.pool._sets.values() for rs in sets) == copies_before
        # an in-place edit of the board drops the stash: the getters then take a kernel of their own (and see the edit)
        state.board[0, 0] = 1024
        assert np.array(get_observation(state))[0] == np.float32(10 / 16)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_more_live_states_than_the_pool_keeps_take_no_mirror():
    """a set the pool does not keep is referenced by nobody once the call returns, so its pinned mirror
    would go back to torch's host allocator while the launch is still writing it.  Such sets get NO mirror (delta 0, plain
    copy path); the values read back are the same either way.  Sets the pool drops (clear, shape eviction) are parked until an
    event behind their last launch has completed."""
    n = 8
    rng = np.random.default_rng(99)
    resultpool.pool.clear()
    held = [batched_reset(_loop_keys(rng, n)) for _ in range(resultpool.MAX_SETS_PER_KEY + 3)]  # the caller keeps every state
    looked = [resultpool.pool.lookup(s.board) for s in held]
    assert all(hit is not None for hit in looked[:resultpool.MAX_SETS_PER_KEY]) and all(hit is None for hit in looked[resultpool.MAX_SETS_PER_KEY:])
    assert all(hit[0].mirror_delta != 0 and hit[0].mirror is not None for hit in looked[:resultpool.MAX_SETS_PER_KEY])
    keys = _loop_keys(rng, n)
    want = oracle.reset_keyed_i32(np.asarray(keys).astype(np.uint32))
    overflow = batched_reset(keys)  # an un-pooled set: no mirror, results through the plain device-to-host path
    assert resultpool.pool.lookup(overflow.board) is None
    assert np.array_equal(np.array(overflow.board).reshape(n, 16), want)
    pooled_again = held[0]
    assert np.array(pooled_again.board).shape == (n, 4, 4)
    # dropping pooled sets parks them (with their mirrors) behind an event on their stream
    resultpool.pool.clear()
    assert len(resultpool.pool._graveyard) <= resultpool.MAX_SETS_PER_KEY
    torch.cuda.synchronize()
    resultpool.pool._reap()
    assert resultpool.pool._graveyard == []
    del held


@pytest.mark.gpu
def test_host_mirror_waits_for_the_stream_the_move_was_launched_on():
    """A move enqueued on a side stream behind a long-running kernel, read back while ANOTHER stream is current: the NumPy read
    synchronises the stream the launch went to (the set is bound to it), so it never sees a half-written mirror."""
    n = 8
    rng = np.random.default_rng(55)
    resultpool.pool.clear()
    side = torch.cuda.Stream()
    big = torch.empty((1 << 26,), dtype=torch.float32, device='cuda')
    for rep in range(5):
        with torch.cuda.stream(side):
            states = batched_reset(_loop_keys(rng, n))
            keys = _loop_keys(rng, n)
            before = np.array(states.board).reshape(n, 16)
            for _ in range(6):
                big.normal_()  # ~1 ms of work on the side stream in front of the move
            new_states, rewards, dones, _ = batched_step(states, np.zeros(n, dtype=np.int32), keys)
        got = np.array(new_states.board).reshape(n, 16), np.array(rewards)  # the default stream is current here
        want = oracle.step_state_i32_keyed(before, np.zeros(n, dtype=np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint8),
                                           np.zeros(n, np.float32), keys)
        assert np.array_equal(got[0], want['boards']) and np.array_equal(got[1], want['reward']), rep
