"""
SURVEY 8(f) N4 -- device-side trajectory recording, episode segmentation, batched normalised reward, on-device
evaluation statistics.  Each test runs on the oracle-backed stand-in (host logic, CPU) and on the GPU with the HIP
library.  What is restated from the reference: the per-move recording and the cut into Trajectory objects of
reinforce/training/self_play.py:359-392, normalize_reward (twentyfortyeight/utils/normalize.py:6-33) and the report of
evaluate_games (self_play.py:503-521).
"""

import numpy as np
import pytest
import torch

import oracle
from oracle_backend import OracleLibrary
from simulate_2048_amd import BatchedEnv2048, capi
from simulate_2048_amd.trajectory import TrajectoryBuffer


@pytest.fixture(params=['host-logic', pytest.param('hip', marks=pytest.mark.gpu)])
def make_env(request):  # noqa: D103
    if request.param == 'hip':
        return lambda n, seed, **kw: BatchedEnv2048(n, seed, device='cuda:0', **kw)
    return lambda n, seed, **kw: BatchedEnv2048(n, seed, device='cpu', lib=OracleLibrary(), **kw)


def host(t):
    return t.cpu().numpy()


def test_step_takes_actions_of_any_integer_dtype_and_device(make_env):
    """A policy's int64 argmax, int32 actions, a host tensor, a strided view: converted into the env's persistent uint8 buffer
    (values mod 4, like the uint8 form) -- the same games as with uint8 actions on the device."""
    n, seed = 513, 11
    a, b = make_env(n, seed, auto_reset=True), make_env(n, seed, auto_reset=True)
    a.reset()
    b.reset()
    rng = np.random.default_rng(2)
    for t in range(12):
        raw = rng.integers(-8, 300, size=n)
        want = torch.from_numpy((raw & 3).astype(np.uint8)).to(a.device)
        a.step(want)
        form = t % 4
        if form == 0:
            given = torch.from_numpy(raw.astype(np.int64)).to(b.device)
        elif form == 1:
            given = torch.from_numpy(raw.astype(np.int32)).to(b.device)
        elif form == 2:
            given = torch.from_numpy(raw.astype(np.int64))  # on the host
        else:
            given = torch.from_numpy(np.stack([raw, raw], axis=1).astype(np.int16)).to(b.device)[:, 0]  # not contiguous
        b.step(given)
        assert torch.equal(a.boards, b.boards) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), t


def test_recorded_rows_match_oracle(make_env):
    """Every row the step kernel writes into the [T, N] arrays -- incl. max_exp (pre-reset) and, with
    normalize_reward, the float32 log-normalised reward (<= 1 ulp from the oracle's double-precision value)."""
    n, T, seed = 700, 40, 5
    for normalize in (False, True):
        env = make_env(n, seed, env_base=3, auto_reset=True, with_legal=True, with_obs=True)
        env.reset()
        env.rollout(150)  # mid/late game: games end inside the recording
        traj = TrajectoryBuffer(env, T, record_policy=True, record_value=True, normalize_reward=normalize)
        traj.begin()
        ref = host(env.boards).copy()
        rng = np.random.default_rng(1)
        finished = 0
        for t in range(T):
            acts = torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).to(env.device)
            pol = torch.from_numpy(rng.random((n, 4)).astype(np.float32)).to(env.device)
            val = torch.from_numpy(rng.random(n).astype(np.float32)).to(env.device)
            traj.step(acts, policy=pol, value=val)
            r = oracle.step_batch_ex(ref, host(acts), seed, 150 + t, env_base=3, flags=1)
            ref = r['boards']
            finished += int(r['done'].sum())
            assert np.array_equal(host(traj.done[t]), r['done']) and np.array_equal(host(traj.max_exp[t]), r['max_exp'])
            assert np.array_equal(host(traj.legal[t + 1]), r['legal']) and np.array_equal(host(traj.obs[t + 1]), r['obs'])
            assert np.array_equal(host(traj.actions[t]), host(acts))
            assert torch.equal(traj.policy[t], pol) and torch.equal(traj.value[t], val)
            if normalize:
                got, want = host(traj.reward[t]), r['reward_norm']
                assert got.dtype == np.float32
                assert np.all(np.abs(got - want) <= np.spacing(np.maximum(want, np.float32(1e-30)))), 'normalised reward: > 1 ulp'
                assert np.all((got == 0) == (r['reward'] == 0))
            else:
                assert np.array_equal(host(traj.reward[t]), r['reward'])
        assert np.array_equal(host(env.boards), ref) and finished > 20
        # a finished game's max_exp row is its LAST board's max tile, not the fresh board's (which is <= 2)
        ended = host(traj.done) == 1
        assert host(traj.max_exp)[ended].min() >= 4


def reference_style_cut(obs, actions, rewards, dones, max_steps):
    """self_play.py:359-392 restated over recorded rows: per game, rows while active; active &= ~done."""
    n = dones.shape[1]
    lists = [dict(obs=[], act=[], rew=[]) for _ in range(n)]
    active = np.ones(n, dtype=bool)
    for t in range(max_steps):
        if not active.any():
            break
        for i in range(n):
            if active[i]:
                lists[i]['obs'].append(obs[t, i])
                lists[i]['act'].append(int(actions[t, i]))
                lists[i]['rew'].append(float(rewards[t, i]))
        active = active & ~dones[t].astype(bool)
    return lists, active


def test_frozen_episodes_equal_the_reference_cut(make_env):
    """No auto-reset (the reference's loop): one episode per game, rows after the first done excluded, Trajectory fields."""
    n, T, seed = 96, 700, 77
    env = make_env(n, seed, with_legal=True, with_obs=True)
    env.reset()
    traj = TrajectoryBuffer(env, T, record_boards=True)
    traj.begin()
    rng = np.random.default_rng(2)
    for t in range(T):
        legal = host(traj.legal[t])
        acts = np.array([rng.choice([a for a in range(4) if (m >> a) & 1]) if m else 0 for m in legal], dtype=np.uint8)
        traj.step(torch.from_numpy(acts).to(env.device))
    ep = traj.episodes(want_episode_of=True)
    lists, still_active = reference_style_cut(host(traj.obs), host(traj.actions), host(traj.reward), host(traj.done), T)
    assert not still_active.any()  # random-legal play ends within 700 moves
    assert len(ep['env']) == n and host(ep['env']).tolist() == list(range(n)) and bool(ep['terminated'].all())
    final_boards = host(traj.boards[T])
    for i in (0, 1, 17, n - 1):
        tr = traj.trajectory(ep, i)
        assert len(lists[i]['act']) == int(ep['length'][i]) == tr['actions'].shape[0]
        assert np.array_equal(host(tr['observations']), np.stack(lists[i]['obs']))
        assert host(tr['actions']).tolist() == lists[i]['act'] and host(tr['rewards']).astype(float).tolist() == lists[i]['rew']
        assert tr['done'] is True and tr['total_reward'] == float(sum(lists[i]['rew']))
        assert tr['max_tile'] == 1 << int(final_boards[i].max())  # frozen games keep their last board
    # ... and all games at once, in the reference's Trajectory layout (host arrays, one copy per row array)
    exported = traj.export_trajectories()
    assert len(exported) == n and all(t.done for t in exported)
    for i in (0, 3, n - 1):
        t = exported[i]
        assert len(t) == len(lists[i]['act']) and t.observations.dtype == np.float32 and t.actions.dtype == np.int32
        assert np.array_equal(t.observations, np.stack(lists[i]['obs'])) and t.actions.tolist() == lists[i]['act']
        assert t.rewards.dtype == np.float32 and t.rewards.tolist() == lists[i]['rew'] and t.total_reward == float(sum(lists[i]['rew']))
        assert t.max_tile == 1 << int(final_boards[i].max()) and t.policies.shape == (len(t), 4) and t.values.shape == (len(t),)
    lengths = host(ep['length'])
    eo = host(ep['episode_of'])
    for i in (0, 5, n - 1):
        assert (eo[: lengths[i], i] == i).all() and (eo[lengths[i]:, i] == -1).all()
    # the same through the oracle's restatement
    want = oracle.segment_episodes(host(traj.done), host(traj.reward), host(traj.max_exp), flags=1)
    assert np.array_equal(host(ep['length']), want['length']) and np.array_equal(host(ep['total_reward']), want['total_reward'])
    assert np.array_equal(host(ep['max_tile']), np.where(want['max_exp'] > 0, 1 << want['max_exp'].astype(np.int64), 0))


def test_auto_reset_episodes_and_ring_mode(make_env):
    n, T, C, seed = 300, 260, 64, 9
    plain_env = make_env(n, seed, auto_reset=True)
    ring_env = make_env(n, seed, auto_reset=True)
    for e in (plain_env, ring_env):
        e.reset()
    plain = TrajectoryBuffer(plain_env, T, record_obs=False, record_legal=False)
    ring = TrajectoryBuffer(ring_env, C, record_obs=False, record_legal=False, ring=True)
    plain.begin()
    ring.begin()
    for t in range(T):
        plain.step(None)
        ring.step(None)
    ep = plain.episodes(want_episode_of=True)
    want, want_of = oracle.segment_episodes(host(plain.done), host(plain.reward), host(plain.max_exp), want_episode_of=True)
    for key in ('env', 'length', 'terminated', 'total_reward'):
        assert np.array_equal(host(ep[key]).astype(want[key].dtype), want[key]), key
    assert np.array_equal(host(ep['start']), want['start']) and np.array_equal(host(ep['episode_of']), want_of)
    assert np.array_equal(host(ep['max_tile']), 1 << want['max_exp'].astype(np.int64))
    # every environment's episodes tile its column exactly
    lengths = np.bincount(want['env'], weights=want['length'], minlength=n)
    assert (lengths == T).all() and len(want) > n and int(want['terminated'].sum()) == int(host(plain.done).sum())
    # ring: rows wrapped 4 times; the window is the last C steps (no obs/legal rows recorded -> all C are valid)
    assert ring.generation == T // C and ring.window() == (T - C, C)
    order = [(T - C + r) % C for r in range(C)]
    assert np.array_equal(host(ring.done)[order], host(plain.done)[T - C:]) and np.array_equal(host(ring.reward)[order], host(plain.reward)[T - C:])
    rep = ring.episodes()
    rwant = oracle.segment_episodes(host(plain.done)[T - C:], host(plain.reward)[T - C:], host(plain.max_exp)[T - C:])
    assert np.array_equal(host(rep['length']), rwant['length']) and np.array_equal(host(rep['start']), rwant['start'].astype(np.int64) + T - C)
    assert np.array_equal(host(rep['total_reward']), rwant['total_reward'])
    with pytest.raises(IndexError):
        plain.step(None)  # a plain buffer is full after `horizon` steps
    # ring with observation rows: the oldest step's "before" row has been reused -> window of C - 1 steps
    ring2 = TrajectoryBuffer(ring_env, 8, ring=True)
    ring2.begin()
    for _ in range(20):
        ring2.step(None)
    assert ring2.window() == (13, 7) and len(ring2.episodes()['env']) >= n


def test_rollout_episode_statistics(make_env):
    """env2048_rollout_episodes: per-episode statistics accumulated on device == the oracle's, games spanning calls are
    counted whole, and the derived report has the keys of evaluate_games (self_play.py:503-521)."""
    n, seed = 900, 31
    env = make_env(n, seed, env_base=10, auto_reset=True)
    env.reset()
    start = host(env.boards).copy()
    stats = env.rollout_episodes(170)
    stats = env.rollout_episodes(230, stats)  # second call continues the games in progress
    ref = oracle.rollout_episodes(start, seed, 0, 400, env_base=10, flags=3)
    assert np.array_equal(host(stats).view(np.uint64), ref['stats'])
    assert np.array_equal(host(env.boards), ref['boards'])
    assert np.array_equal(host(env._ep_length).astype(np.uint32), ref['ep_length']) and np.array_equal(host(env._ep_return), ref['ep_return'])
    rep = BatchedEnv2048.episode_statistics(stats)
    w = ref['stats']
    games = int(w[0])
    assert rep['games'] == games > n and rep['mean_length'] == int(w[1]) / games and rep['mean_reward'] == int(w[2]) / games
    assert 100 < rep['mean_length'] < 200 and rep['min_reward'] <= rep['mean_reward'] <= rep['max_reward']
    assert sum(rep['max_tile_histogram'].values()) == games and rep['max_tile'] in (128, 256, 512)
    assert set(rep) >= {'mean_reward', 'std_reward', 'max_reward', 'min_reward', 'mean_max_tile', 'max_tile', 'mean_length',
                        'reached_2048', 'reached_4096', 'reached_8192', 'reached_16384', 'reached_32768'}
    # the plain rollout is untouched by the statistics variant: same boards from the same start
    plain = make_env(n, seed, env_base=10, auto_reset=True)
    plain.reset()
    plain.rollout(400)
    assert torch.equal(plain.boards, env.boards)


def test_random_play_distribution_matches_the_reference_engine(make_env, golden):
    """Distribution-level parity of the STOCHASTIC half (the reference's spawn bitstream cannot be reproduced, its distribution
    can): 3000 whole games of uniformly random play by the reference engine with its own NumPy RNG
    (tests/golden/random_play_stats.npz, oracle/gen_golden.py) against first games of this build's Philox-driven play --
    mean game length, mean total reward (z-tests, |z| < 4.5 with the reference sample's standard error) and the max-tile
    histogram (chi-square of the reference counts against this build's shares, p > 1e-4).  Also exercises the
    no-auto-reset form of env2048_rollout_episodes: every environment's game is counted exactly once."""
    from scipy import stats as scipy_stats

    g = golden['random_play_stats']
    n = 40_000
    env = make_env(n, 20481, auto_reset=False)
    env.reset()
    block = env.rollout_episodes(900)
    block = env.rollout_episodes(900, block)  # long enough for every game to end; finished games are not counted twice
    rep = BatchedEnv2048.episode_statistics(block)
    assert rep['games'] == n and int((env._ep_length == -1).sum()) == n  # the ENV2048_EPISODE_OVER marker, as int32
    ref_len, ref_rew = g['length'].astype(np.float64), g['total_reward'].astype(np.float64)
    z_len = (rep['mean_length'] - ref_len.mean()) / (ref_len.std(ddof=1) / np.sqrt(len(ref_len)))
    z_rew = (rep['mean_reward'] - ref_rew.mean()) / (ref_rew.std(ddof=1) / np.sqrt(len(ref_rew)))
    assert abs(z_len) < 4.5 and abs(z_rew) < 4.5, (z_len, z_rew, rep['mean_length'], ref_len.mean())
    assert abs(rep['std_length'] / ref_len.std(ddof=1) - 1.0) < 0.08 and abs(rep['std_reward'] / ref_rew.std(ddof=1) - 1.0) < 0.08
    tiles = [32, 64, 128, 256]  # the bins with enough reference games; the rest pooled
    ours = np.array([rep['max_tile_histogram'].get(t, 0) for t in tiles], dtype=np.float64)
    ours = np.append(ours, n - ours.sum()) / n
    theirs = np.array([(g['max_tile'] == t).sum() for t in tiles], dtype=np.float64)
    theirs = np.append(theirs, len(g['max_tile']) - theirs.sum())
    chi2 = scipy_stats.chisquare(theirs, ours * theirs.sum())
    assert chi2.pvalue > 1e-4, (chi2, ours, theirs / theirs.sum())


def test_replay_equals_the_same_steps_one_by_one(make_env):
    """env2048_replay: T pre-staged moves in one launch == T step calls -- boards, reward rows, done rows, per-env sums;
    ragged batch, odd env_base, with and without auto-reset, optional outputs left out."""
    n, T, seed = 1500, 37, 13
    rng = np.random.default_rng(3)
    for auto_reset in (True, False):
        a = make_env(n, seed, env_base=5, auto_reset=auto_reset)
        b = make_env(n, seed, env_base=5, auto_reset=auto_reset)
        for e in (a, b):
            e.reset()
            e.rollout(120)  # late game: games end during the replay
        actions = torch.from_numpy(rng.integers(0, 4, size=(T, n)).astype(np.uint8)).to(a.device)
        rew = torch.zeros((T, n), dtype=torch.int32, device=a.device)
        don = torch.zeros((T, n), dtype=torch.uint8, device=a.device)
        for t in range(T):
            a.step(actions[t], out_reward=rew[t], out_done=don[t])
        rew2 = torch.full((T, n), -1, dtype=torch.int32, device=a.device)
        don2 = torch.full((T, n), 9, dtype=torch.uint8, device=a.device)
        out = b.replay(actions, out_reward=rew2, out_done=don2)
        assert torch.equal(a.boards, b.boards) and torch.equal(rew, rew2) and torch.equal(don, don2)
        assert a.step_index == b.step_index
        assert torch.equal(out['return_sum'], rew.sum(dim=0, dtype=torch.int64)) and torch.equal(out['episodes'].long(), don.long().sum(dim=0))
        assert int(out['valid_moves'].sum()) > 0 and int(don.sum()) > (20 if auto_reset else 5)
        c = make_env(n, seed, env_base=5, auto_reset=auto_reset)
        c.reset()
        c.rollout(120)
        c.replay(actions)  # no row outputs
        assert torch.equal(c.boards, a.boards)
    with pytest.raises(ValueError):
        b.replay(actions[:, :10].contiguous())
    if a.device.type == 'cuda':  # both tilings of the rollout / replay / statistics kernels give the same results
        lib, s = a.lib, torch.cuda.current_stream().cuda_stream
        results = []
        for tile in (capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE):
            boards = c.boards.clone()
            rr = torch.zeros((T, n), dtype=torch.int32, device='cuda')
            lib.replay(0, s, n, 5, boards.data_ptr(), actions.data_ptr(), rr.data_ptr(), None, n, seed, 0, T, tile, None, None, None)
            st = torch.zeros((40,), dtype=torch.int64, device='cuda')
            st[4] = -1
            b2 = c.boards.clone()
            lib.rollout_episodes(0, s, n, 5, b2.data_ptr(), seed, 0, 300, 3 | tile, None, None, st.data_ptr())
            torch.cuda.synchronize()
            results.append((boards, rr, b2, st))
        for x, y in zip(*results):
            assert torch.equal(x, y)


@pytest.mark.filterwarnings('ignore:The CUDA Graph is empty')  # the capture that is refused on purpose below
def test_device_step_counter_and_graph_replay(make_env):
    """Records with ``step_counter``: the launch's step index is device counter + offset, so (a) stepping with the counter ==
    stepping with explicit indices, and (b) on the GPU a captured hipGraph of K step launches + env2048_advance_counter(K)
    can be replayed and CONTINUES the random stream (each replay == the next K explicit steps)."""
    n, seed = 3000, 21
    ref = make_env(n, seed, env_base=9, auto_reset=True, with_legal=True)
    env = make_env(n, seed, env_base=9, auto_reset=True, with_legal=True, device_step_counter=True)
    for e in (ref, env):
        e.reset()
        e.rollout(100)
    assert env.step_index == ref.step_index == 100
    rng = np.random.default_rng(8)
    acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).to(env.device) for _ in range(4)]
    for t in range(12):
        ref.step(acts[t & 3])
        env.step(acts[t & 3])
    assert env.step_index == 112 and int(env.step_counter[0]) == 112  # eager steps advance the device counter themselves
    env.commit_steps()  # nothing pending: a no-op
    assert env.step_index == 112 and int(env.step_counter[0]) == 112
    assert torch.equal(env.boards, ref.boards) and torch.equal(env.legal, ref.legal) and torch.equal(env.reward, ref.reward)
    env.step_index = 500  # re-synchronise both sides somewhere else
    ref.step_index = 500
    env.step(None)
    ref.step(None)
    assert torch.equal(env.boards, ref.boards) and env.step_index == 501
    if env.device.type != 'cuda':
        return
    K = 6
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):  # a captured region that ends WITHOUT commit_steps(): its replays would repeat step indices ...
        with torch.cuda.graph(torch.cuda.CUDAGraph(), stream=side):
            env.step(acts[0])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='commit_steps'):  # ... which the next call reports (and then starts clean)
        env.step(acts[0])
    with torch.cuda.stream(side):
        with torch.cuda.graph(torch.cuda.CUDAGraph(), stream=side):
            env.step(acts[0])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='INSIDE'):  # committing after the capture block is too late as well
        env.commit_steps()
    torch.cuda.synchronize()
    assert env.step_index == 501
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for t in range(K):
                env.step(acts[t & 3])
            env.commit_steps()  # the graph's last node: counter += K
    torch.cuda.synchronize()
    assert env.step_index == 501  # capture launched nothing
    for replay in range(3):
        graph.replay()
        for t in range(K):
            ref.step(acts[t & 3])
        torch.cuda.synchronize()
        assert torch.equal(env.boards, ref.boards) and torch.equal(env.done, ref.done), f'replay {replay}'
        assert env.step_index == 501 + K * (replay + 1) == ref.step_index


@pytest.mark.gpu
@pytest.mark.parametrize('n,stride', [(304, 304), (301, 304), (300, 301), (4, 4), (3, 8), (1024, 1028), (4100, 4100)])
@pytest.mark.parametrize('flags', [0, 1, 2, 3])
def test_episode_kernels_on_random_rows(n, stride, flags):
    """env2048_count_episodes / env2048_segment_episodes on random rows of a ring, against the oracle: batches the
    four-games-per-lane form takes (n and row_stride multiples of 4, aligned rows) and ones it does not (odd n, odd stride,
    rows at an odd offset), frozen games and float rewards, with and without episode_of."""
    lib = capi.library()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(n * 7 + stride + flags)
    ring, rows, first = 50, 37, 20
    done = (rng.random((ring, stride)) < 0.07).astype(np.uint8)
    reward = rng.integers(0, 5000, size=(ring, stride)).astype(np.int32)
    if flags & 2:
        reward = (reward.astype(np.float32) / 7.0).astype(np.float32)
    max_exp = rng.integers(1, 16, size=(ring, stride)).astype(np.uint8)
    want, want_of = oracle.segment_episodes(done, reward, max_exp, n=n, rows=rows, first_row=first, ring_rows=ring, flags=flags, want_episode_of=True)
    for shift in (0, 1):  # (shift 1: every row starts one element later -- the aligned form cannot take it)
        pad = lambda a: torch.from_numpy(np.concatenate([np.zeros(shift, dtype=a.dtype), a.reshape(-1)])).cuda()[shift:]
        d_done, d_reward, d_max = pad(done), pad(reward), pad(max_exp)
        counts = torch.zeros((n,), dtype=torch.int32, device='cuda')
        lib.count_episodes(0, s, n, stride, rows, first, ring, d_done.data_ptr(), flags & 1, counts.data_ptr())
        ends = torch.cumsum(counts.to(torch.int64), dim=0)
        offsets = (ends - counts).contiguous()
        assert int(ends[-1]) == len(want)
        raw = torch.zeros((max(len(want), 1), 24), dtype=torch.uint8, device='cuda')
        episode_of = torch.full((ring * stride + shift,), -7, dtype=torch.int32, device='cuda')[shift:]
        lib.segment_episodes(0, s, n, stride, rows, first, ring, d_done.data_ptr(), d_reward.data_ptr(), d_max.data_ptr(), flags, offsets.data_ptr(),
                             raw.data_ptr(), episode_of.data_ptr())
        got = raw.cpu().numpy()[:len(want)].view(oracle.EPISODE_DTYPE).reshape(-1)
        for field in ('env', 'start', 'length', 'terminated', 'max_exp', 'total_reward'):
            np.testing.assert_array_equal(got[field], want[field], err_msg=f'{field} shift={shift}')
        np.testing.assert_array_equal(episode_of.cpu().numpy().reshape(ring, stride), want_of)
        lib.segment_episodes(0, s, n, stride, rows, first, ring, d_done.data_ptr(), None, None, flags & 1, offsets.data_ptr(), raw.data_ptr(), None)
        got = raw.cpu().numpy()[:len(want)].view(oracle.EPISODE_DTYPE).reshape(-1)
        np.testing.assert_array_equal(got['length'], want['length'])
        assert (got['total_reward'] == 0).all()


def test_episode_entry_points_reject_bad_arguments(make_env):
    env = make_env(8, 1)
    if isinstance(env.lib, OracleLibrary):
        pytest.skip('argument validation lives in the C-ABI')
    done = torch.zeros((4, 8), dtype=torch.uint8, device=env.device)
    counts = torch.zeros((8,), dtype=torch.int32, device=env.device)
    s = torch.cuda.current_stream().cuda_stream
    with pytest.raises(capi.Env2048Error) as e:
        env.lib.count_episodes(0, s, 8, 7, 4, 0, 4, done.data_ptr(), 0, counts.data_ptr())  # stride < n
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        env.lib.count_episodes(0, s, 8, 8, 5, 0, 4, done.data_ptr(), 0, counts.data_ptr())  # more rows than the ring holds
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        env.lib.count_episodes(0, s, 8, 8, 4, 0, 4, done.data_ptr(), 8, counts.data_ptr())  # unknown flag
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(ValueError):
        env.step(None, out_reward=torch.zeros(8, dtype=torch.int32, device=env.device), normalized_reward=True)


# ---------------------------------------------------------------------------------------------------------------------------
# Round 3: ADVICE r2 fixes (episode carry across reset, random-action rows) and the staged-moves entry point
# ---------------------------------------------------------------------------------------------------------------------------
def test_rollout_episodes_counts_again_after_reset(make_env):
    """reset -> rollout_episodes -> reset -> rollout_episodes: the second run reports its own games.  Without auto-reset
    every environment's game is counted once and marked over; round 2 kept those marks across reset() and then reported
    0 games; with auto-reset it carried stale lengths into the first game after a reset."""
    n, seed = 256, 4
    env = make_env(n, seed, auto_reset=False)
    env.reset()
    first = BatchedEnv2048.episode_statistics(env.rollout_episodes(3000))
    assert first['games'] == n
    env.reset(step_index=0)
    again = BatchedEnv2048.episode_statistics(env.rollout_episodes(3000))
    assert again == first  # same seed, same step indices, fresh carry: the very same games
    auto = make_env(n, seed, auto_reset=True)
    auto.reset()
    auto.rollout_episodes(10)  # no game can end within 10 moves (a board fills up after 14 at the earliest): every slot carries 10
    assert int(host(auto._ep_length).min()) == 10
    auto.reset(step_index=0)
    assert int(host(auto._ep_length).max()) == 0
    stats = BatchedEnv2048.episode_statistics(auto.rollout_episodes(600))
    oracle_stats = oracle.rollout_episodes(oracle.reset_batch(n, seed, 0)[0], seed, 0, 600, flags=3)['stats']
    assert stats == BatchedEnv2048.episode_statistics(torch.from_numpy(oracle_stats.astype(np.int64)))
    auto.set_boards(host(auto.boards))  # overwriting the state forgets the games in progress as well
    assert int(host(auto._ep_length).max()) == 0


def test_random_action_steps_record_their_moves(make_env):
    """traj.step(None): the action row holds the moves the on-device random policy played (low two bits of the step words),
    and the recorded step is bit for bit the step(None) of an unrecorded env."""
    n, T, seed, base = 300, 12, 77, 5
    env = make_env(n, seed, env_base=base, auto_reset=True)
    plain = make_env(n, seed, env_base=base, auto_reset=True)
    for e in (env, plain):
        e.reset()
        e.rollout(30)
    traj = TrajectoryBuffer(env, T, record_obs=True, record_legal=False)
    traj.begin()
    for t in range(T):
        traj.step(None)
        plain.step(None)
        want = np.array([oracle.step_word(seed, base + i, 30 + t) & 3 for i in range(n)], dtype=np.uint8)
        assert np.array_equal(host(traj.actions[t]), want)
        assert np.array_equal(host(env.boards), host(plain.boards)) and np.array_equal(host(traj.reward[t]), host(plain.reward))
    games = traj.export_trajectories()
    assert games and all(g.actions.max() <= 3 for g in games) and any(g.actions.any() for g in games)
    counter_env = make_env(n, seed, env_base=base, auto_reset=True, device_step_counter=True)
    counter_env.reset()
    t2 = TrajectoryBuffer(counter_env, 4, record_obs=True, record_legal=False)
    t2.begin()
    t2.step(None)
    with pytest.raises(RuntimeError, match='action rows'):
        t2.export_trajectories()


@pytest.mark.parametrize('n', [1000, 5000])
def test_step_many_staged_moves_equal_single_steps(make_env, n):
    """BatchedEnv2048.step_many(actions[K, N]): one env2048_replay launch ('replay', the default for small batches), K
    launches from one env2048_step_sequence call ('launches'), and K step() calls give the same boards and rows."""
    K, seed = 9, 31
    rng = np.random.default_rng(n)
    actions = torch.from_numpy(rng.integers(0, 4, size=(K, n)).astype(np.uint8))
    envs = {mode: make_env(n, seed, env_base=7, auto_reset=True) for mode in ('steps', 'auto', 'replay', 'launches')}
    for e in envs.values():
        e.reset()
        e.rollout(120)
    actions = actions.to(envs['steps'].device)
    rows = {}
    ref_r = torch.zeros((K, n), dtype=torch.int32, device=actions.device)
    ref_d = torch.zeros((K, n), dtype=torch.uint8, device=actions.device)
    for k in range(K):
        envs['steps'].step(actions[k], out_reward=ref_r[k], out_done=ref_d[k])
    for mode in ('auto', 'replay', 'launches'):
        calls = getattr(envs[mode].lib, 'calls', None)
        mark = len(calls) if calls is not None else 0
        _, r, d = envs[mode].step_many(actions, mode=mode)
        rows[mode] = (r, d)
        assert envs[mode].step_index == envs['steps'].step_index
        assert torch.equal(envs[mode].boards, envs['steps'].boards) and torch.equal(r, ref_r) and torch.equal(d, ref_d)
        # the persistent reward / done describe the last move, as after step()
        assert torch.equal(envs[mode].reward, ref_r[K - 1]) and torch.equal(envs[mode].done, ref_d[K - 1])
        if calls is not None:  # host logic: which entry point was used
            names = [c[0] for c in calls[mark:]]
            assert names == (['replay'] if mode in ('auto', 'replay') else ['step_sequence'] if 'step_sequence' in names else names)
            assert ('replay' in names) == (mode in ('auto', 'replay'))
    with pytest.raises(ValueError):
        envs['auto'].step_many(actions[:, :10].contiguous())
    obs_env = make_env(n, seed, auto_reset=True, with_obs=True)
    obs_env.reset()
    with pytest.raises(ValueError, match='replay'):
        obs_env.step_many(actions, mode='replay')
    obs_env.step_many(actions)  # 'auto' falls back to launches and refreshes the persistent obs
    st = obs_env.status()
    assert torch.equal(st['obs'], obs_env.obs)
