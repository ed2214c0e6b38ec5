"""
GPU tests (-m gpu) that run every BASELINE.json config at ITS OWN size through the C-ABI:

* configs[1]  65 536 parallel boards, mid-game (reset + 64 random steps), host uint8 actions      -> vs oracle, every step
* configs[2]  1 048 576 boards, host actions (the headline)                                       -> vs the OpenMP oracle
* configs[4]  1 048 576 boards x 512-step rollout, Philox spawn, auto-reset: ONE env2048_rollout == 512 env2048_step
              launches (enqueued by env2048_step_sequence, rewards/done recorded as [512, 2^20] rows) -- boards, sum of
              rewards, episode counts; a 4096-environment slice against the oracle; SURVEY App. D sanity statistics
              (mean episode length 142-160 steps, ~16 % invalid moves)
* configs[3]  (8 x 2^20 boards over 8 GPUs) is covered on one GPU by test_hip_parity.test_config4_slicing_at_8x2pow20 and on
              two gloo ranks by test_host_logic; configs[0] (one board, 1000 steps) by test_api_twentyfortyeight.

plus the entry points that enqueue several launches per call (env2048_step_sequence / env2048_step_many).
Caller loop these replace: reinforce/training/self_play.py:314-368 (reference).
"""

import ctypes

import numpy as np
import pytest
import torch

import oracle
from simulate_2048_amd import capi

pytestmark = pytest.mark.gpu

DEV = 0
SEED = 0x2048
AUTO, RANDOM = capi.FLAG_AUTO_RESET, capi.FLAG_RANDOM_ACTIONS


@pytest.fixture(scope='module')
def lib():
    L = capi.library()
    assert L.device_count() >= 1
    return L


def stream():
    return torch.cuda.current_stream(DEV).cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def midgame(lib, n, env_base=0):
    """configs[1]/[2] board generation: env2048_reset(seed=0x2048, step 0) then 64 random steps."""
    b = torch.empty((n, 16), dtype=torch.uint8, device='cuda')
    d = torch.empty((n,), dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), n, env_base, p(b), p(d), SEED, 0)
    lib.rollout(DEV, stream(), n, env_base, p(b), SEED, 0, 64, None, None, None, AUTO | RANDOM)
    torch.cuda.synchronize()
    return b


@pytest.mark.parametrize('n', [1 << 16, 1 << 20])
def test_configs_1_and_2_host_actions_vs_oracle(lib, n):
    steps = 12 if n == 1 << 16 else 4
    b = midgame(lib, n)
    ref = oracle.reset_batch(n, SEED, 0, threads=16)[0]
    ref = oracle.rollout_batch(ref, SEED, 0, 64, flags=AUTO | RANDOM, threads=16)['boards']
    assert np.array_equal(b.cpu().numpy(), ref), 'mid-game boards differ'
    assert 3 < float((ref > 0).sum(axis=1).mean()) < 14  # mid-game: neither fresh nor full
    gen = torch.Generator(device='cpu').manual_seed(1)
    reward = torch.empty((n,), dtype=torch.int32, device='cuda')
    done = torch.empty((n,), dtype=torch.uint8, device='cuda')
    legal = torch.empty((n,), dtype=torch.uint8, device='cuda')
    obs = torch.empty((n, 16), dtype=torch.float32, device='cuda')
    for t in range(64, 64 + steps):
        actions = torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8)
        a = actions.cuda()
        extras = (t & 1) == 0  # alternate the lean and the legal+obs kernel
        lib.step(DEV, stream(), n, 0, p(b), p(a), p(reward), p(done), p(legal) if extras else None, p(obs) if extras else None,
                 SEED, t, AUTO)
        r = oracle.step_batch(ref, actions.numpy(), SEED, t, flags=AUTO, want_legal=extras, want_obs=extras, threads=16)
        ref = r['boards']
        torch.cuda.synchronize()
        assert np.array_equal(b.cpu().numpy(), ref), f'boards differ at step {t}'
        assert np.array_equal(reward.cpu().numpy(), r['reward']), f'reward differs at step {t}'
        assert np.array_equal(done.cpu().numpy(), r['done']), f'done differs at step {t}'
        if extras:
            assert np.array_equal(legal.cpu().numpy(), r['legal']), f'legal differs at step {t}'
            assert np.array_equal(obs.cpu().numpy(), r['obs']), f'obs differs at step {t}'


def test_config_4_rollout_512_equals_512_steps_and_oracle(lib):
    n, T = 1 << 20, 512
    flags = AUTO | RANDOM
    start = torch.empty((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), n, 0, p(start), None, SEED, 0)

    # (a) 512 env2048_step launches, rewards / done flags recorded as rows of [T, n] arrays by the kernel itself
    stepped = start.clone()
    reward_rows = torch.full((T, n), -1, dtype=torch.int32, device='cuda')
    done_rows = torch.full((T, n), 7, dtype=torch.uint8, device='cuda')
    lib.step_sequence(DEV, stream(), n, 0, p(stepped), None, p(reward_rows), p(done_rows), None, None, n, SEED, 0, T, flags)
    # (b) one fused rollout
    rolled = start.clone()
    ret = torch.zeros((n,), dtype=torch.int64, device='cuda')
    eps = torch.zeros((n,), dtype=torch.int32, device='cuda')
    moves = torch.zeros((n,), dtype=torch.int32, device='cuda')
    lib.rollout(DEV, stream(), n, 0, p(rolled), SEED, 0, T, p(ret), p(eps), p(moves), flags)
    torch.cuda.synchronize()
    assert int(stepped.max()) < 32
    assert torch.equal(stepped, rolled), 'boards after 512 steps != boards after rollout(512)'
    assert torch.equal(reward_rows.sum(dim=0, dtype=torch.int64), ret), 'per-environment reward sums differ'
    assert torch.equal(done_rows.sum(dim=0, dtype=torch.int64), eps.long()), 'per-environment episode counts differ'
    assert int(done_rows.max()) <= 1 and int(reward_rows.min()) >= 0

    # (c) a 4096-environment slice of the same job on the CPU oracle (global indices lo .. lo+4095)
    lo, m = 300_000, 4096
    ref = oracle.rollout_batch(start[lo:lo + m].cpu().numpy(), SEED, 0, T, env_base=lo, flags=flags, threads=16)
    assert np.array_equal(rolled[lo:lo + m].cpu().numpy(), ref['boards'])
    assert np.array_equal(ret[lo:lo + m].cpu().numpy(), ref['return_sum'])
    assert np.array_equal(eps[lo:lo + m].cpu().numpy().astype(np.uint32), ref['episodes'])
    assert np.array_equal(moves[lo:lo + m].cpu().numpy().astype(np.uint32), ref['valid_moves'])

    # (d) sanity against the reference's random-play statistics (300 reference games gave a mean
    #     length of 141.7 moves incl. invalid ones and 16.4 % invalid moves)
    total = n * T
    episodes = int(eps.sum(dtype=torch.int64))
    mean_len = total / episodes
    invalid = 1.0 - int(moves.sum(dtype=torch.int64)) / total
    assert 130.0 < mean_len < 170.0, mean_len
    assert 0.13 < invalid < 0.20, invalid


def test_replay_at_full_size_equals_step_sequence(lib):
    """2^20 boards x 64 staged moves: one env2048_replay launch == 64 env2048_step launches (boards and every row)."""
    n, T = 1 << 20, 64
    start = midgame(lib, n)
    gen = torch.Generator(device='cpu').manual_seed(5)
    actions = torch.randint(0, 4, (T, n), generator=gen, dtype=torch.uint8).cuda()
    table = (ctypes.c_void_p * T)(*[actions[t].data_ptr() for t in range(T)])
    b1, b2 = start.clone(), start.clone()
    r1 = torch.zeros((T, n), dtype=torch.int32, device='cuda')
    d1 = torch.zeros((T, n), dtype=torch.uint8, device='cuda')
    r2, d2 = torch.full_like(r1, -1), torch.full_like(d1, 9)
    lib.step_sequence(DEV, stream(), n, 0, p(b1), table, p(r1), p(d1), None, None, n, SEED, 64, T, AUTO)
    ret = torch.zeros((n,), dtype=torch.int64, device='cuda')
    eps = torch.zeros((n,), dtype=torch.int32, device='cuda')
    lib.replay(DEV, stream(), n, 0, p(b2), p(actions), p(r2), p(d2), n, SEED, 64, T, AUTO, p(ret), p(eps), None)
    torch.cuda.synchronize()
    assert torch.equal(b1, b2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert torch.equal(ret, r1.sum(dim=0, dtype=torch.int64)) and torch.equal(eps.long(), d1.long().sum(dim=0))
    with pytest.raises(capi.Env2048Error) as e:
        lib.replay(DEV, stream(), n, 0, p(b2), p(actions), None, None, n - 1, SEED, 0, T, AUTO, None, None, None)
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.replay(DEV, stream(), n, 0, p(b2), p(actions), None, None, n, SEED, 0, T, RANDOM, None, None, None)  # actions are given
    assert e.value.code == capi.ERR_INVALID_ARG


def test_both_tilings_agree_at_full_size_and_around_the_switch(lib):
    """The step kernel's two tilings (1 / 4 boards per lane; the launch size picks one) give identical boards, rewards,
    done flags, legal masks and observations: at 2^20 + 77 boards (ragged in both), and on either side of the 2^18-board
    switch with the default policy against the oracle."""
    n = (1 << 20) + 77
    start = midgame(lib, n, env_base=2)
    a = torch.randint(0, 4, (n,), dtype=torch.uint8, device='cuda')
    outs = []
    for tile in (capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE):
        b = start.clone()
        r = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        lg = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        ob = torch.zeros((n, 16), dtype=torch.float32, device='cuda')
        for t in range(3):
            lib.step(DEV, stream(), n, 2, p(b), p(a), p(r), p(d), p(lg), p(ob), SEED, 64 + t, AUTO | tile)
        lib.step(DEV, stream(), n, 2, p(b), p(a), p(r), p(d), None, None, SEED, 67, AUTO | tile)  # the lean kernel too
        torch.cuda.synchronize()
        outs.append((b, r, d, lg, ob))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    for n in ((1 << 18) - 3, 1 << 18, (1 << 18) + 1, (1 << 18) + 2049):
        start = midgame(lib, n)
        acts = torch.randint(0, 4, (n,), dtype=torch.uint8)
        b = start.clone()
        r = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        a = acts.cuda()
        lib.step(DEV, stream(), n, 0, p(b), p(a), p(r), p(d), None, None, SEED, 64, AUTO)
        ref = oracle.step_batch(start.cpu().numpy(), acts.numpy(), SEED, 64, flags=AUTO, threads=16)
        torch.cuda.synchronize()
        assert np.array_equal(b.cpu().numpy(), ref['boards']) and np.array_equal(r.cpu().numpy(), ref['reward'])
        assert np.array_equal(d.cpu().numpy(), ref['done'])
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(DEV, stream(), 8, 0, p(b), p(a), p(r), p(d), None, None, 0, 0, capi.FLAG_TILE_SMALL | capi.FLAG_TILE_LARGE)
    assert e.value.code == capi.ERR_INVALID_ARG


def test_step_sequence_rows_equal_single_steps(lib):
    """env2048_step_sequence == the same env2048_step calls one by one: every output row, with host action
    tables, row_stride 0 (overwrite) and > n (padded rows), f32 and bf16 observations, a ragged batch."""
    n, T, stride = 5000, 9, 5120
    rng = np.random.default_rng(5)
    start = midgame(lib, n, env_base=3)
    acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for _ in range(T)]
    table = (ctypes.c_void_p * T)(*[a.data_ptr() for a in acts])
    for obs_flag, obs_dtype in ((0, torch.float32), (capi.FLAG_OBS_BF16, torch.bfloat16)):
        flags = AUTO | obs_flag
        # one by one
        b1 = start.clone()
        r1 = torch.zeros((T, n), dtype=torch.int32, device='cuda')
        d1 = torch.zeros((T, n), dtype=torch.uint8, device='cuda')
        l1 = torch.zeros((T, n), dtype=torch.uint8, device='cuda')
        o1 = torch.zeros((T, n, 16), dtype=obs_dtype, device='cuda')
        for t in range(T):
            lib.step(DEV, stream(), n, 3, p(b1), p(acts[t]), p(r1[t]), p(d1[t]), p(l1[t]), p(o1[t]), 99, 1000 + t, flags)
        # one call, padded rows
        b2 = start.clone()
        r2 = torch.full((T, stride), -5, dtype=torch.int32, device='cuda')
        d2 = torch.full((T, stride), 9, dtype=torch.uint8, device='cuda')
        l2 = torch.full((T, stride), 9, dtype=torch.uint8, device='cuda')
        o2 = torch.full((T, stride, 16), -1.0, dtype=obs_dtype, device='cuda')
        lib.step_sequence(DEV, stream(), n, 3, p(b2), table, p(r2), p(d2), p(l2), p(o2), stride, 99, 1000, T, flags)
        torch.cuda.synchronize()
        assert torch.equal(b1, b2)
        assert torch.equal(r1, r2[:, :n]) and torch.equal(d1, d2[:, :n]) and torch.equal(l1, l2[:, :n])
        assert torch.equal(o1, o2[:, :n])
        assert bool((r2[:, n:] == -5).all()) and bool((d2[:, n:] == 9).all()) and bool((o2[:, n:] == -1.0).all())  # padding untouched
        # row_stride 0: every step overwrites the same arrays -> the last step's outputs remain
        b3 = start.clone()
        r3 = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d3 = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        lib.step_sequence(DEV, stream(), n, 3, p(b3), table, p(r3), p(d3), None, None, 0, 99, 1000, T, flags)
        torch.cuda.synchronize()
        assert torch.equal(b3, b1) and torch.equal(r3, r1[T - 1]) and torch.equal(d3, d1[T - 1])
    # argument checks: nothing is enqueued by a failing call
    before = b3.clone()
    with pytest.raises(capi.Env2048Error) as e:
        lib.step_sequence(DEV, stream(), n, 3, p(b3), None, p(r3), p(d3), None, None, 0, 99, 0, T, AUTO)  # no actions, not RANDOM
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.step_sequence(DEV, stream(), n, 3, p(b3), table, p(r3), p(d3), None, None, n - 1, 99, 0, T, AUTO)  # overlapping rows
    assert e.value.code == capi.ERR_INVALID_ARG
    torch.cuda.synchronize()
    assert torch.equal(b3, before)
    lib.step_sequence(DEV, stream(), n, 3, p(b3), table, p(r3), p(d3), None, None, 0, 99, 0, 0, AUTO)  # count 0: no-op


def test_step_many_equals_single_steps_and_store_policy_flags(lib):
    """env2048_step_many over several independent batches == one env2048_step per record; the STORES_* hints
    change no result; a bad record anywhere in the array enqueues nothing."""
    rng = np.random.default_rng(6)
    sizes = [4096, 2048 * 3 + 17, 1, 70_000]
    bases = [0, 10_001, 7, 1 << 34]
    starts = [midgame(lib, n, env_base=base) for n, base in zip(sizes, bases)]
    acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for n in sizes]
    policies = [0, capi.FLAG_STORES_WRITE_THROUGH, capi.FLAG_STORES_WRITE_BACK, 0]
    want = []
    for n, base, s, a in zip(sizes, bases, starts, acts):
        b = s.clone()
        r = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        lib.step(DEV, stream(), n, base, p(b), p(a), p(r), p(d), None, None, 5, 77, AUTO)
        want.append((b, r, d))
    got = [(s.clone(), torch.zeros((n,), dtype=torch.int32, device='cuda'), torch.zeros((n,), dtype=torch.uint8, device='cuda'))
           for n, s in zip(sizes, starts)]
    records = (capi.StepArgs * len(sizes))()
    for k, (n, base, a, (b, r, d)) in enumerate(zip(sizes, bases, acts, got)):
        records[k] = capi.StepArgs(n, base, p(b), p(a), p(r), p(d), None, None, 5, 77, AUTO | policies[k], 0)
    lib.step_many(DEV, stream(), records)
    torch.cuda.synchronize()
    for (b, r, d), (wb, wr, wd) in zip(got, want):
        assert torch.equal(b, wb) and torch.equal(r, wr) and torch.equal(d, wd)
    # an invalid record (misaligned boards) at the END: the valid ones before it must not have run
    snapshot = [b.clone() for b, _, _ in got]
    records[len(sizes) - 1].boards = p(got[-1][0]) + 1
    with pytest.raises(capi.Env2048Error) as e:
        lib.step_many(DEV, stream(), records)
    assert e.value.code == capi.ERR_ALIGNMENT
    torch.cuda.synchronize()
    for (b, _, _), s in zip(got, snapshot):
        assert torch.equal(b, s)
    with pytest.raises(capi.Env2048Error) as e:  # both store hints at once
        lib.step(DEV, stream(), 1, 0, p(got[2][0]), p(acts[2]), None, p(got[2][2]), None, None, 0, 0,
                 capi.FLAG_STORES_WRITE_THROUGH | capi.FLAG_STORES_WRITE_BACK)
    assert e.value.code == capi.ERR_INVALID_ARG
