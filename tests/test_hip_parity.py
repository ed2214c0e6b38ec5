"""
GPU parity tests (-m gpu): the HIP path, called through the C-ABI (capi.Library), against

* the golden fixtures generated from the reference NumPy engine (tests/golden), bit for bit;
* the CPU oracle on seeded random inputs at sizes it finishes in seconds, bit for bit;
* size-independent properties at BASELINE.json's full sizes (2^20 boards): slicing invariance,
  rollout == T single steps, tile-mass conservation, checksums against the oracle.

Everything here is integer / byte / index work: the bar is bit-exact (obs floats included: they
are exact multiples of 1/16).
"""

import numpy as np
import pytest
import torch

import oracle
from simulate_2048_amd import capi

pytestmark = pytest.mark.gpu

DEV = 0


@pytest.fixture(scope='module')
def lib():
    L = capi.library()
    assert L.device_count() >= 1
    return L


def cu(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda(DEV).contiguous()


def stream():
    return torch.cuda.current_stream(DEV).cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def hip_step(lib, boards, actions, seed, step_index, *, env_base=0, flags=0, extras=False):
    """Runs the launch in BOTH tilings of the step kernel (ENV2048_FLAG_TILE_SMALL / _LARGE) and insists on identical
    outputs, so every oracle comparison in this file covers both; returns the default-policy result."""
    outs = [_hip_step(lib, boards, actions, seed, step_index, env_base=env_base, flags=flags | tile, extras=extras)
            for tile in (0, capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE)]
    for other in outs[1:]:
        for k in outs[0]:
            np.testing.assert_array_equal(other[k], outs[0][k], err_msg=f'tilings differ in {k}')
    return outs[0]


def _hip_step(lib, boards, actions, seed, step_index, *, env_base=0, flags=0, extras=False):
    n = boards.shape[0]
    b = cu(boards)
    a = None if actions is None else cu(actions, torch.uint8)
    reward = torch.full((n,), -7, dtype=torch.int32, device='cuda')
    done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    legal = torch.full((n,), 255, dtype=torch.uint8, device='cuda') if extras else None
    obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda') if extras else None
    lib.step(DEV, stream(), n, env_base, p(b), p(a), p(reward), p(done), p(legal), p(obs), seed, step_index, flags)
    torch.cuda.synchronize()
    out = {'boards': b.cpu().numpy(), 'reward': reward.cpu().numpy(), 'done': done.cpu().numpy()}
    if extras:
        out['legal'] = legal.cpu().numpy()
        out['obs'] = obs.cpu().numpy()
    return out


def hip_afterstate(lib, boards, actions):
    n = boards.shape[0]
    b = cu(boards)
    out = torch.zeros_like(b)
    reward = torch.zeros((n,), dtype=torch.int32, device='cuda')
    changed = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    a = cu(actions, torch.uint8)  # keep every device buffer referenced until the kernel has run
    lib.afterstate(DEV, stream(), n, p(b), p(a), p(out), p(reward), p(changed))
    torch.cuda.synchronize()
    return {'boards': out.cpu().numpy(), 'reward': reward.cpu().numpy(), 'changed': changed.cpu().numpy()}


def hip_status(lib, boards):
    n = boards.shape[0]
    b = cu(boards)
    done = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    legal = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    obs = torch.zeros((n, 16), dtype=torch.float32, device='cuda')
    max_exp = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    n_empty = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.status(DEV, stream(), n, p(b), p(done), p(legal), p(obs), p(max_exp), p(n_empty))
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in dict(done=done, legal=legal, obs=obs, max_exp=max_exp, n_empty=n_empty).items()}


def hip_spawn(lib, boards, k, e):
    n = boards.shape[0]
    b = cu(boards)
    placed = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    kk, ee = cu(k, torch.uint8), cu(e, torch.uint8)
    lib.spawn(DEV, stream(), n, p(b), p(kk), p(ee), p(placed))
    torch.cuda.synchronize()
    return {'boards': b.cpu().numpy(), 'placed': placed.cpu().numpy()}


def hip_reset(lib, n, seed, step_index, env_base=0):
    b = torch.full((n, 16), 77, dtype=torch.uint8, device='cuda')
    done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), n, env_base, p(b), p(done), seed, step_index)
    torch.cuda.synchronize()
    return b.cpu().numpy(), done.cpu().numpy()


def hip_rollout(lib, boards, seed, step0, T, env_base=0, flags=0):
    """Both tilings of the rollout kernel (one / four boards per lane) must agree; returns the default-policy result."""
    outs = [_hip_rollout(lib, boards, seed, step0, T, env_base, flags | tile) for tile in (0, capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE)]
    for other in outs[1:]:
        for k in outs[0]:
            np.testing.assert_array_equal(other[k], outs[0][k], err_msg=f'rollout tilings differ in {k}')
    return outs[0]


def _hip_rollout(lib, boards, seed, step0, T, env_base=0, flags=0):
    n = boards.shape[0]
    b = cu(boards)
    ret = torch.zeros((n,), dtype=torch.int64, device='cuda')
    eps = torch.zeros((n,), dtype=torch.int32, device='cuda')
    moves = torch.zeros((n,), dtype=torch.int32, device='cuda')
    lib.rollout(DEV, stream(), n, env_base, p(b), seed, step0, T, p(ret), p(eps), p(moves), flags)
    torch.cuda.synchronize()
    return {'boards': b.cpu().numpy(), 'return_sum': ret.cpu().numpy(), 'episodes': eps.cpu().numpy().astype(np.uint32),
            'valid_moves': moves.cpu().numpy().astype(np.uint32)}


def random_boards(rng, n):
    """Mixture that hits every branch: sparse, dense/near-terminal, big tiles, hand cases."""
    parts = [
        rng.integers(0, 7, size=(n // 4, 16)),
        np.where(rng.random((n // 4, 16)) < 0.08, 0, rng.integers(1, 5, size=(n // 4, 16))),
        np.where(rng.random((n // 4, 16)) < 0.3, 0, rng.integers(0, 18, size=(n // 4, 16))),
        rng.integers(1, 4, size=(n - 3 * (n // 4), 16)),
    ]
    b = np.concatenate(parts).astype(np.uint8)
    b[0] = 0
    b[1] = np.arange(1, 17)
    b[2] = 16
    b[3] = 17
    return b


# ---------------------------------------------------------------------------- golden fixtures
def test_rows_exhaustive_on_gpu(lib, golden):
    """All 18^4 rows, each as row 0 of a board moved left AND as column 0 moved up, etc."""
    g = golden['rows']
    base = int(g['max_e']) + 1
    idx = np.arange(base**4)
    rows = np.stack([(idx // base**3) % base, (idx // base**2) % base, (idx // base) % base, idx % base], axis=1).astype(np.uint8)
    n = rows.shape[0]
    for action in range(4):
        boards = np.zeros((n, 4, 4), dtype=np.uint8)
        line = rows if action in (0, 1) else rows[:, ::-1]
        want = g['merged'] if action in (0, 1) else g['merged'][:, ::-1]
        if action in (0, 2):
            boards[:, 2, :] = line
        else:
            boards[:, :, 1] = line
        res = hip_afterstate(lib, boards.reshape(n, 16), np.full(n, action, dtype=np.uint8))
        got = res['boards'].reshape(n, 4, 4)
        got_line = got[:, 2, :] if action in (0, 2) else got[:, :, 1]
        np.testing.assert_array_equal(got_line, want)
        np.testing.assert_array_equal(res['reward'], g['score'])
        np.testing.assert_array_equal(res['changed'], g['can_move'])
        assert (got != 0).sum() == (want != 0).sum()  # nothing leaked into the other lines


def test_afterstates_golden(lib, golden):
    g = golden['afterstates']
    boards = g['boards']
    n = boards.shape[0]
    for a in range(4):
        res = hip_afterstate(lib, boards, np.full(n, a, dtype=np.uint8))
        np.testing.assert_array_equal(res['boards'], g['after'][:, a])
        np.testing.assert_array_equal(res['reward'], g['reward'][:, a])
        np.testing.assert_array_equal(res['changed'], g['valid'][:, a])
    # all four afterstates in one launch (the reference's latent_state per action / legal_actions_mask)
    b = cu(boards)
    out4 = torch.full((n, 4, 16), 99, dtype=torch.uint8, device='cuda')
    reward4 = torch.full((n, 4), -1, dtype=torch.int32, device='cuda')
    legal4 = torch.full((n,), 99, dtype=torch.uint8, device='cuda')
    lib.afterstates4(DEV, stream(), n, p(b), p(out4), p(reward4), p(legal4))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out4.cpu().numpy(), g['after'])
    np.testing.assert_array_equal(reward4.cpu().numpy(), g['reward'])
    np.testing.assert_array_equal(legal4.cpu().numpy(), g['legal'])
    st = hip_status(lib, boards)
    np.testing.assert_array_equal(st['legal'], g['legal'])
    np.testing.assert_array_equal(st['done'], g['done'])
    np.testing.assert_array_equal(st['n_empty'], (boards == 0).sum(axis=1))
    np.testing.assert_array_equal(st['max_exp'], boards.max(axis=1))
    np.testing.assert_array_equal(st['obs'], boards.astype(np.float32) / 16.0)


def test_next_state_golden(lib, golden):
    """The reference's REAL next_state outputs, replayed with its spawn choice injected."""
    g = golden['next_state']
    after = hip_afterstate(lib, g['boards'], g['actions'])
    moved = g['spawn_k'] != 255
    np.testing.assert_array_equal(after['changed'].astype(bool), moved)
    k = np.where(moved, g['spawn_k'], 200).astype(np.uint8)
    out = hip_spawn(lib, after['boards'], k, g['spawn_e'])
    np.testing.assert_array_equal(out['boards'], g['out'])
    np.testing.assert_array_equal(np.where(moved, after['reward'], 0), g['reward'])
    np.testing.assert_array_equal(hip_status(lib, out['boards'])['done'], g['done'])
    ref = oracle.spawn_batch(after['boards'], k, g['spawn_e'])
    np.testing.assert_array_equal(out['placed'], ref['placed'])


@pytest.mark.parametrize('name', ['auto', 'plain'])
def test_trajectories_golden(lib, golden, name):
    g = golden['trajectories']
    seed, env_base, step0 = int(g[f'{name}_seed']), int(g[f'{name}_env_base']), int(g[f'{name}_step0'])
    flags = capi.FLAG_AUTO_RESET if int(g[f'{name}_auto_reset']) else 0
    T, n = g[f'{name}_actions'].shape
    boards, done = hip_reset(lib, n, seed, step0, env_base)
    np.testing.assert_array_equal(boards, g[f'{name}_boards0'])
    assert not done.any()
    for t in range(T):
        res = hip_step(lib, boards, g[f'{name}_actions'][t], seed, step0 + t, env_base=env_base, flags=flags, extras=True)
        boards = res['boards']
        np.testing.assert_array_equal(boards, g[f'{name}_boards'][t], err_msg=f't={t}')
        np.testing.assert_array_equal(res['reward'], g[f'{name}_reward'][t])
        np.testing.assert_array_equal(res['done'], g[f'{name}_done'][t])
        np.testing.assert_array_equal(res['legal'], g[f'{name}_legal'][t])
        np.testing.assert_array_equal(res['obs'], boards.astype(np.float32) / 16.0)
    ro = hip_rollout(lib, g[f'{name}_boards0'], seed, step0, T, env_base, flags)
    np.testing.assert_array_equal(ro['boards'], boards)
    np.testing.assert_array_equal(ro['return_sum'], g[f'{name}_reward'].sum(axis=0))
    np.testing.assert_array_equal(ro['episodes'], g[f'{name}_done'].sum(axis=0))


def test_resets_golden(lib, golden):
    g = golden['resets']
    boards, done = hip_reset(lib, g['boards'].shape[0], int(g['seed']), int(g['step']), int(g['env_base']))
    np.testing.assert_array_equal(boards, g['boards'])
    assert not done.any()


# ---------------------------------------------------------------------------- seeded vs oracle
@pytest.mark.parametrize('flags', [0, capi.FLAG_AUTO_RESET, capi.FLAG_RANDOM_ACTIONS,
                                   capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS])
def test_step_matches_oracle_random_boards(lib, flags):
    rng = np.random.default_rng(1 + flags)
    n = 50_003  # ragged: not a multiple of the block or wave size
    boards = random_boards(rng, n)
    actions = rng.integers(0, 256, size=n).astype(np.uint8)  # any byte: taken mod 4
    seed, step, env_base = 0xDEADBEEFCAFEF00D, (1 << 40) + 3, (1 << 33) + 11
    acts = None if flags & capi.FLAG_RANDOM_ACTIONS else actions
    got = hip_step(lib, boards, acts, seed, step, env_base=env_base, flags=flags, extras=True)
    ref = oracle.step_batch(boards, acts, seed, step, env_base=env_base, flags=flags, want_legal=True, want_obs=True,
                            threads=8)
    for k in ('boards', 'reward', 'done', 'legal', 'obs'):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    lean = hip_step(lib, boards, acts, seed, step, env_base=env_base, flags=flags, extras=False)
    for k in ('boards', 'reward', 'done'):
        np.testing.assert_array_equal(lean[k], ref[k], err_msg='lean ' + k)


@pytest.mark.parametrize('n', [1, 3, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096, 6145])
def test_step_tile_and_block_edges(lib, n):
    """Batch sizes around the wave tile (256 boards) and block (2048) boundaries, every alignment of the
    first environment within its Philox group (env_base & 3), near-terminal boards so that the reset
    queue is exercised, all four kernel variants."""
    rng = np.random.default_rng(1000 + n)
    boards = np.where(rng.random((n, 16)) < 0.06, 0, rng.integers(1, 4, size=(n, 16))).astype(np.uint8)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    for env_base in (0, 1, 2, 3, (1 << 32) - 2):
        for flags in (capi.FLAG_AUTO_RESET, capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS, 0):
            acts = None if flags & capi.FLAG_RANDOM_ACTIONS else actions
            ref = oracle.step_batch(boards, acts, 77, 5, env_base=env_base, flags=flags, want_legal=True, want_obs=True)
            for extras in (False, True):
                got = hip_step(lib, boards, acts, 77, 5, env_base=env_base, flags=flags, extras=extras)
                for k in ('boards', 'reward', 'done') + (('legal', 'obs') if extras else ()):
                    np.testing.assert_array_equal(got[k], ref[k], err_msg=f'{k} n={n} env_base={env_base} flags={flags}')


@pytest.mark.parametrize('obs_flag,torch_dtype', [(capi.FLAG_OBS_F16, torch.float16), (capi.FLAG_OBS_BF16, torch.bfloat16)])
def test_sixteen_bit_observations(lib, obs_flag, torch_dtype):
    """ENV2048_FLAG_OBS_F16 / _BF16: bit-identical values to the f32 observation (e/16 is exact in all three),
    full blocks (LDS-transposed stores), ragged tail (per-lane stores) and rebuilt boards (reset pass)."""
    rng = np.random.default_rng(31)
    for n in (2048 * 3, 4999):
        boards = np.where(rng.random((n, 16)) < 0.06, 0, rng.integers(1, 4, size=(n, 16))).astype(np.uint8)  # many terminal
        boards[: n // 3] = random_boards(rng, n // 3)  # exponents up to 17
        actions = rng.integers(0, 4, size=n).astype(np.uint8)
        for flags in (capi.FLAG_AUTO_RESET, 0):
            ref = oracle.step_batch(boards, actions, 5, 9, env_base=7, flags=flags, want_legal=True, want_obs=True)
            b, a = cu(boards), cu(actions)
            reward = torch.zeros((n,), dtype=torch.int32, device='cuda')
            done = torch.zeros((n,), dtype=torch.uint8, device='cuda')
            legal = torch.zeros((n,), dtype=torch.uint8, device='cuda')
            obs = torch.full((n, 16), -1.0, dtype=torch_dtype, device='cuda')
            lib.step(DEV, stream(), n, 7, p(b), p(a), p(reward), p(done), p(legal), p(obs), 5, 9, flags | obs_flag)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(b.cpu().numpy(), ref['boards'])
            np.testing.assert_array_equal(legal.cpu().numpy(), ref['legal'])
            np.testing.assert_array_equal(obs.to(torch.float32).cpu().numpy(), ref['obs'], err_msg=f'n={n} flags={flags}')
    with pytest.raises(capi.Env2048Error):
        lib.step(DEV, stream(), n, 7, p(b), p(a), p(reward), p(done), None, p(obs), 5, 9, capi.FLAG_OBS_F16 | capi.FLAG_OBS_BF16)


def test_long_random_play_matches_oracle(lib):
    """4096 games x 300 steps with auto-reset: every branch of real play, compared every step."""
    n, T, seed = 4096, 300, 20480
    boards, _ = hip_reset(lib, n, seed, 0)
    ref_boards, _ = oracle.reset_batch(n, seed, 0, threads=8)
    np.testing.assert_array_equal(boards, ref_boards)
    b = cu(boards)
    reward = torch.zeros((n,), dtype=torch.int32, device='cuda')
    done = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    episodes = 0
    for t in range(T):
        lib.step(DEV, stream(), n, 0, p(b), None, p(reward), p(done), None, None, seed, t, flags)
        ref = oracle.step_batch(ref_boards, None, seed, t, flags=flags, threads=8)
        ref_boards = ref['boards']
        torch.cuda.synchronize()
        np.testing.assert_array_equal(b.cpu().numpy(), ref_boards, err_msg=f't={t}')
        np.testing.assert_array_equal(reward.cpu().numpy(), ref['reward'])
        np.testing.assert_array_equal(done.cpu().numpy(), ref['done'])
        episodes += int(ref['done'].sum())
    assert episodes > n  # every env finished at least one game on average (mean length ~142)


def test_empty_and_single(lib):
    lib.step(DEV, stream(), 0, 0, None, None, None, None, None, None, 1, 2, 0)  # n = 0: no-op, no pointer checks
    lib.reset(DEV, stream(), 0, 0, None, None, 1, 2)
    one = oracle.pack_values([[2, 2, 0, 0], [0] * 4, [0] * 4, [0] * 4]).reshape(1, 16)
    got = hip_step(lib, one, np.array([0], dtype=np.uint8), 42, 0)
    ref = oracle.step_batch(one, [0], 42, 0)
    np.testing.assert_array_equal(got['boards'], ref['boards'])
    assert got['reward'][0] == 4  # tests/test_environment.py:138-149


def test_pack_unpack_roundtrip(lib):
    rng = np.random.default_rng(5)
    n = 10_001
    exps = rng.integers(0, 18, size=(n, 16)).astype(np.uint8)
    for dtype, pack, unpack in ((torch.int32, lib.pack_i32, lib.unpack_i32), (torch.int64, lib.pack_i64, lib.unpack_i64)):
        b = cu(exps)
        vals = torch.zeros((n, 16), dtype=dtype, device='cuda')
        unpack(DEV, stream(), n, p(b), p(vals))
        back = torch.zeros_like(b)
        pack(DEV, stream(), n, p(vals), p(back))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(vals.cpu().numpy(), oracle.unpack_values(exps))
        np.testing.assert_array_equal(back.cpu().numpy(), exps)


# ---------------------------------------------------------------------------- error behaviour
def test_error_codes(lib):
    b = torch.zeros((8, 16), dtype=torch.uint8, device='cuda')
    d = torch.zeros((8,), dtype=torch.uint8, device='cuda')
    a = torch.zeros((8,), dtype=torch.uint8, device='cuda')
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(DEV, stream(), 8, 0, p(b), None, None, p(d), None, None, 0, 0, 0)  # actions NULL without RANDOM_ACTIONS
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(DEV, stream(), 8, 0, p(b) + 4, p(a), None, p(d), None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_ALIGNMENT
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(DEV, stream(), 8, 0, p(b), p(a), None, p(d), None, None, 0, 0, 0x8000)  # unknown flag bit
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(99, stream(), 8, 0, p(b), p(a), None, p(d), None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_DEVICE
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(DEV, stream(), 8, (1 << 63), p(b), p(a), None, p(d), None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_RANGE


# ---------------------------------------------------------------------------- full-size properties
FULL_N = 1 << 20


def test_full_size_slicing_invariance_and_rollout(lib):
    """2^20 boards: two half-batches with the right env_base == the whole batch; the fused
    rollout == T single steps; total tile mass is conserved by every non-reset step."""
    seed, T = 0x2048, 24
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    whole = torch.zeros((FULL_N, 16), dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), FULL_N, 0, p(whole), None, seed, 0)
    start = whole.clone()
    reward = torch.zeros((FULL_N,), dtype=torch.int32, device='cuda')
    done = torch.zeros((FULL_N,), dtype=torch.uint8, device='cuda')
    total = torch.zeros((FULL_N,), dtype=torch.int64, device='cuda')
    episodes = torch.zeros((FULL_N,), dtype=torch.int64, device='cuda')
    pow2 = torch.tensor([0] + [1 << e for e in range(1, 32)], dtype=torch.int64, device='cuda')
    for t in range(T):
        mass_before = pow2[whole.long()].sum(dim=1)
        lib.step(DEV, stream(), FULL_N, 0, p(whole), None, p(reward), p(done), None, None, seed, t, flags)
        assert int(whole.max()) < 32, 'corrupt board bytes'  # (before they are used as an index below)
        mass_after = pow2[whole.long()].sum(dim=1)
        grew = mass_after - mass_before
        live = done == 0
        # a valid move adds exactly one tile of value 2 or 4; an invalid one adds nothing
        assert bool(((grew[live] == 0) | (grew[live] == 2) | (grew[live] == 4)).all())
        total += reward
        episodes += done
    # halves
    half = FULL_N // 2
    for lo in (0, half):
        part = start[lo:lo + half].clone()
        r2 = torch.zeros((half,), dtype=torch.int32, device='cuda')
        d2 = torch.zeros((half,), dtype=torch.uint8, device='cuda')
        for t in range(T):
            lib.step(DEV, stream(), half, lo, p(part), None, p(r2), p(d2), None, None, seed, t, flags)
        assert torch.equal(part, whole[lo:lo + half])
    # fused rollout
    ro = start.clone()
    ret = torch.zeros((FULL_N,), dtype=torch.int64, device='cuda')
    eps = torch.zeros((FULL_N,), dtype=torch.int32, device='cuda')
    lib.rollout(DEV, stream(), FULL_N, 0, p(ro), seed, 0, T, p(ret), p(eps), None, flags)
    torch.cuda.synchronize()
    assert torch.equal(ro, whole)
    assert torch.equal(ret, total)
    assert torch.equal(eps.long(), episodes)


def test_full_size_checksum_vs_oracle(lib):
    """2^20 boards x 3 steps: byte-exact against the (OpenMP) oracle."""
    seed = 77
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    ref_boards, _ = oracle.reset_batch(FULL_N, seed, 0, threads=16)
    b = torch.zeros((FULL_N, 16), dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), FULL_N, 0, p(b), None, seed, 0)
    reward = torch.zeros((FULL_N,), dtype=torch.int32, device='cuda')
    done = torch.zeros((FULL_N,), dtype=torch.uint8, device='cuda')
    # warm the boards into mid-game with the fused rollout on both sides
    lib.rollout(DEV, stream(), FULL_N, 0, p(b), seed, 0, 100, None, None, None, flags)
    ref_boards = oracle.rollout_batch(ref_boards, seed, 0, 100, flags=flags, threads=16)['boards']
    for t in range(100, 103):
        lib.step(DEV, stream(), FULL_N, 0, p(b), None, p(reward), p(done), None, None, seed, t, flags)
        ref = oracle.step_batch(ref_boards, None, seed, t, flags=flags, threads=16)
        ref_boards = ref['boards']
        torch.cuda.synchronize()
        assert np.array_equal(b.cpu().numpy(), ref_boards)
        assert np.array_equal(reward.cpu().numpy(), ref['reward'])
        assert np.array_equal(done.cpu().numpy(), ref['done'])


# ---------------------------------------------------------------------------- entry points added for the API mirrors
def test_spawn_and_status_every_occupancy(lib):
    """Every occupancy pattern of a board: k-th empty cell for every rank (tiles 2 and 4, neighbours up to
    exponent 31) and done / legal / counts, HIP vs oracle."""
    rng = np.random.default_rng(21)
    occ = (np.arange(1 << 16, dtype=np.uint32)[:, None] >> np.arange(16, dtype=np.uint32)) & 1
    base = (occ * rng.integers(1, 32, size=occ.shape)).astype(np.uint8)
    boards = np.tile(base, (17, 1))
    k = np.repeat(np.arange(17, dtype=np.uint8), base.shape[0])
    e = rng.integers(1, 3, size=boards.shape[0]).astype(np.uint8)
    ref = oracle.spawn_batch(boards, k, e)
    got = hip_spawn(lib, boards, k, e)
    np.testing.assert_array_equal(got['boards'], ref['boards'])
    np.testing.assert_array_equal(got['placed'], ref['placed'])
    for lo, hi in ((1, 2), (30, 31), (1, 31)):
        b = (occ * rng.integers(lo, hi + 1, size=occ.shape)).astype(np.uint8)
        ref = oracle.status_batch(b)
        got = hip_status(lib, b)
        for key in ('done', 'legal', 'n_empty', 'max_exp'):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f'{key} [{lo},{hi}]')
        np.testing.assert_array_equal(got['obs'], ref['obs'])


def test_afterstates4_matches_oracle_random_boards(lib):
    rng = np.random.default_rng(41)
    n = 30_011
    boards = random_boards(rng, n)
    b = cu(boards)
    out4 = torch.zeros((n, 4, 16), dtype=torch.uint8, device='cuda')
    reward4 = torch.zeros((n, 4), dtype=torch.int32, device='cuda')
    legal4 = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.afterstates4(DEV, stream(), n, p(b), p(out4), p(reward4), p(legal4))
    torch.cuda.synchronize()
    legal = np.zeros(n, dtype=np.uint8)
    for a in range(4):
        ref = oracle.afterstate_batch(boards, np.full(n, a, dtype=np.uint8))
        np.testing.assert_array_equal(out4.cpu().numpy()[:, a], ref['boards'], err_msg=f'action {a}')
        np.testing.assert_array_equal(reward4.cpu().numpy()[:, a], ref['reward'])
        legal |= (ref['changed'] << a).astype(np.uint8)
    np.testing.assert_array_equal(legal4.cpu().numpy(), legal)
    np.testing.assert_array_equal(legal, oracle.status_batch(boards)['legal'])
    np.testing.assert_array_equal(b.cpu().numpy(), boards)  # input untouched


def test_chance_outcomes_match_oracle(lib, golden):
    boards = golden['afterstates']['boards']
    n = boards.shape[0]
    b = cu(boards)
    out = torch.zeros((n, 32, 16), dtype=torch.uint8, device='cuda')
    prob = torch.zeros((n, 32), dtype=torch.float32, device='cuda')
    count = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.chance_outcomes(DEV, stream(), n, p(b), p(out), p(prob), p(count))
    torch.cuda.synchronize()
    ref = oracle.chance_outcomes(boards)
    np.testing.assert_array_equal(out.cpu().numpy(), ref['boards'])
    np.testing.assert_array_equal(prob.cpu().numpy(), ref['prob'])  # same f32 division on both sides
    np.testing.assert_array_equal(count.cpu().numpy(), ref['count'])
    sums = prob.sum(dim=1).cpu().numpy()
    np.testing.assert_allclose(sums, 1.0, atol=1e-6)


def test_chance_outcomes_vs_reference_after_state(lib, golden):
    """env2048_chance_outcomes against the reference's own after_state outputs (tests/golden/chance_outcomes.npz):
    order and boards bit-exact; probabilities are float32 on device vs float64 in the reference: 2 ulp f32."""
    g = golden['chance_outcomes']
    n = g['boards'].shape[0]
    b = cu(g['boards'])
    out = torch.full((n, 32, 16), 99, dtype=torch.uint8, device='cuda')
    prob = torch.full((n, 32), -1.0, dtype=torch.float32, device='cuda')
    count = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.chance_outcomes(DEV, stream(), n, p(b), p(out), p(prob), p(count))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(count.cpu().numpy(), g['count'])
    np.testing.assert_array_equal(out.cpu().numpy(), g['outcomes'])
    np.testing.assert_allclose(prob.cpu().numpy().astype(np.float64), g['prob'], rtol=2.4e-7, atol=0.0)


def test_expectimax_ply_matches_oracle_and_the_unfused_chain(lib, golden):
    """env2048_expectimax_ply == oracle, == env2048_afterstates4 -> env2048_chance_outcomes -> env2048_status composed by hand,
    and its afterstate scores == the reference's latent_state scores (golden); ragged sizes around the 64-board wave tile."""
    all_boards = np.concatenate([golden['afterstates']['boards'], golden['chance_outcomes']['boards']])
    for n in (1, 63, 64, 65, 257, all_boards.shape[0]):
        boards = all_boards[:n]
        b = cu(boards)
        sb = torch.full((n, 4, 32, 16), 99, dtype=torch.uint8, device='cuda')
        sp = torch.full((n, 4, 32), -1.0, dtype=torch.float32, device='cuda')
        ss = torch.full((n, 4, 32), 99, dtype=torch.uint8, device='cuda')
        sc = torch.full((n, 4), 99, dtype=torch.uint8, device='cuda')
        ar = torch.full((n, 4), -1, dtype=torch.int32, device='cuda')
        lg = torch.full((n,), 99, dtype=torch.uint8, device='cuda')
        lib.expectimax_ply(DEV, stream(), n, p(b), p(sb), p(sp), p(ss), p(sc), p(ar), p(lg))
        torch.cuda.synchronize()
        ref = oracle.expectimax_ply(boards)
        for got, key in ((sb, 'boards'), (sp, 'prob'), (ss, 'status'), (sc, 'count'), (ar, 'reward'), (lg, 'legal')):
            np.testing.assert_array_equal(got.cpu().numpy(), ref[key], err_msg=f'{key} n={n}')
        np.testing.assert_array_equal(b.cpu().numpy(), boards)  # input untouched
        # optional outputs left out
        sb2 = torch.zeros_like(sb)
        sp2 = torch.zeros_like(sp)
        lib.expectimax_ply(DEV, stream(), n, p(b), p(sb2), p(sp2), None, None, None, None)
        torch.cuda.synchronize()
        assert torch.equal(sb2, sb) and torch.equal(sp2, sp)
    # the unfused chain on the GPU
    n = all_boards.shape[0]
    a4 = torch.zeros((n, 4, 16), dtype=torch.uint8, device='cuda')
    r4 = torch.zeros((n, 4), dtype=torch.int32, device='cuda')
    l4 = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.afterstates4(DEV, stream(), n, p(b), p(a4), p(r4), p(l4))
    ob = torch.zeros((4 * n, 32, 16), dtype=torch.uint8, device='cuda')
    op = torch.zeros((4 * n, 32), dtype=torch.float32, device='cuda')
    lib.chance_outcomes(DEV, stream(), 4 * n, p(a4), p(ob), p(op), None)
    st_legal = torch.zeros((4 * n * 32,), dtype=torch.uint8, device='cuda')
    st_done = torch.zeros((4 * n * 32,), dtype=torch.uint8, device='cuda')
    lib.status(DEV, stream(), 4 * n * 32, p(ob), p(st_done), p(st_legal), None, None, None)
    torch.cuda.synchronize()
    is_legal = ((l4[:, None] >> torch.arange(4, device='cuda', dtype=torch.uint8)) & 1).bool()  # (n, 4)
    mask = is_legal[:, :, None]
    assert torch.equal(torch.where(mask[..., None], ob.reshape(n, 4, 32, 16), torch.zeros_like(sb)), sb)
    assert torch.equal(torch.where(mask, op.reshape(n, 4, 32), torch.zeros_like(sp)), sp)
    used = sp > 0
    chain_status = (st_legal | (st_done << 4)).reshape(n, 4, 32)
    assert torch.equal(torch.where(used, chain_status, torch.zeros_like(ss)), ss)
    assert torch.equal(r4, ar) and torch.equal(l4, lg)
    m = golden['afterstates']['boards'].shape[0]
    np.testing.assert_array_equal(ar.cpu().numpy()[:m], golden['afterstates']['reward'])  # the reference's own scores
    sums = sp.sum(dim=2)
    assert torch.allclose(sums[is_legal], torch.ones_like(sums[is_legal]), atol=1e-6) and bool((sums[~is_legal] == 0).all())


def test_spawn_random_and_onehot_match_oracle(lib, golden):
    boards = golden['afterstates']['boards']
    n = boards.shape[0]
    for count, env_base in ((1, 0), (2, 1001), (3, (1 << 33) + 2)):
        b = cu(boards)
        lib.spawn_random(DEV, stream(), n, env_base, p(b), 99, 7, count)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(b.cpu().numpy(), oracle.spawn_random_batch(boards, 99, 7, count, env_base=env_base))
    small = np.minimum(boards, 30)
    b = cu(small)
    out = torch.zeros((n, 16 * 31), dtype=torch.uint8, device='cuda')
    lib.encode_onehot(DEV, stream(), n, p(b), 31, p(out))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.encode_onehot(small, 31))
    g = golden['transforms']
    b = cu(g['boards'])
    out = torch.zeros((g['boards'].shape[0], 16 * 31), dtype=torch.uint8, device='cuda')
    lib.encode_onehot(DEV, stream(), g['boards'].shape[0], p(b), 31, p(out))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), g['onehot31'])  # the reference's encode_flatten(..., 31)


@pytest.mark.parametrize('width', [1, 2, 3, 15, 16, 17, 31, 32, 33, 255])
def test_onehot_every_width_class(lib, width):
    """One-hot encoder for widths below / at / above the 16-byte store granule, ragged board count, with a
    16-byte aligned output (wide-store kernel) and a 4-byte aligned one (dword kernel)."""
    rng = np.random.default_rng(width)
    n = 64 * 3 + 7
    boards = rng.integers(0, min(width, 32), size=(n, 16)).astype(np.uint8)
    want = oracle.encode_onehot(boards, width)
    b = cu(boards)
    for shift in (0, 4):  # byte offset of the output inside a 16-byte aligned allocation
        raw = torch.full((n * 16 * width + 16,), 7, dtype=torch.uint8, device='cuda')
        out = raw[shift:shift + n * 16 * width]
        lib.encode_onehot(DEV, stream(), n, p(b), width, out.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().reshape(n, 16 * width), want, err_msg=f'shift={shift}')
        assert int(raw[shift + n * 16 * width:].min()) == 7  # nothing written past the end


def test_trajectory_buffer_on_gpu(lib):
    from simulate_2048_amd import BatchedEnv2048
    from simulate_2048_amd.trajectory import TrajectoryBuffer

    n, T, seed = 5000, 20, 17
    env = BatchedEnv2048(n, seed, device='cuda:0', auto_reset=True)
    env.reset()
    traj = TrajectoryBuffer(env, T, record_boards=True)
    traj.begin()
    ref, _ = oracle.reset_batch(n, seed, 0)
    for t in range(T):
        traj.step(None)
        res = oracle.step_batch(ref, None, seed, t, flags=3, want_legal=True, want_obs=True)
        ref = res['boards']
    torch.cuda.synchronize()
    np.testing.assert_array_equal(traj.boards[T].cpu().numpy(), ref)
    np.testing.assert_array_equal(traj.legal[T].cpu().numpy(), res['legal'])
    np.testing.assert_array_equal(traj.obs[T].cpu().numpy(), res['obs'])
    np.testing.assert_array_equal(traj.reward[T - 1].cpu().numpy(), res['reward'])


def test_keyed_step_and_reset_match_oracle(lib):
    """One PRNG key per environment (vmap semantics): env2048_reset_keyed_i32 / env2048_step_state_i32_keyed vs the
    oracle, and row i == the scalar-seed entry point with seed (k0 << 32) | k1 on a batch of one."""
    rng = np.random.default_rng(12)
    n = 20_003
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    keys[0] = (0, 0)
    keys[1] = (0xFFFFFFFF, 0xFFFFFFFF)
    k = cu(keys.view(np.int32))
    fresh = torch.full((n, 16), -1, dtype=torch.int32, device='cuda')
    lib.reset_keyed_i32(DEV, stream(), n, p(k), p(fresh))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(fresh.cpu().numpy(), oracle.reset_keyed_i32(keys))
    values = oracle.unpack_values(random_boards(rng, n)).astype(np.int32)
    actions = rng.integers(-3, 9, size=n).astype(np.int32)
    step_count = rng.integers(0, 500, size=n).astype(np.int32)
    done = (rng.random(n) < 0.2).astype(np.uint8)
    total = rng.integers(0, 5000, size=n).astype(np.float32)
    ref = oracle.step_state_i32_keyed(values, actions, step_count, done, total, keys)
    b_in, a, sc_in, d_in, tr_in = cu(values), cu(actions), cu(step_count), cu(done), cu(total)
    b_out = torch.full((n, 16), -5, dtype=torch.int32, device='cuda')
    sc_out = torch.full((n,), -5, dtype=torch.int32, device='cuda')
    d_out = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    tr_out = torch.full((n,), -5.0, dtype=torch.float32, device='cuda')
    reward = torch.full((n,), -1.0, dtype=torch.float32, device='cuda')
    max_tile = torch.full((n,), -1, dtype=torch.int32, device='cuda')
    lib.step_state_i32_keyed(DEV, stream(), n, p(k), p(b_in), p(b_out), p(a), p(sc_in), p(sc_out), p(d_in), p(d_out), p(tr_in),
                             p(tr_out), p(reward), p(max_tile))
    torch.cuda.synchronize()
    for got, key in ((b_out, 'boards'), (sc_out, 'step_count'), (d_out, 'done'), (tr_out, 'total_reward'), (reward, 'reward'),
                     (max_tile, 'max_tile')):
        np.testing.assert_array_equal(got.cpu().numpy(), ref[key], err_msg=key)
    np.testing.assert_array_equal(b_in.cpu().numpy(), values)  # input only read
    for i in (0, 1, 777, n - 1):  # the scalar-seed form on a batch of one gives the same row
        seed = (int(keys[i, 0]) << 32) | int(keys[i, 1])
        one = oracle.step_state_i32(values[i:i + 1], actions[i:i + 1], step_count[i:i + 1], done[i:i + 1], total[i:i + 1], seed, 0)
        np.testing.assert_array_equal(one['boards'][0], ref['boards'][i])
    with pytest.raises(capi.Env2048Error) as e:
        lib.step_state_i32_keyed(DEV, stream(), n, None, p(b_in), p(b_out), p(a), None, None, p(d_in), p(d_out), None, None,
                                 p(reward), None)
    assert e.value.code == capi.ERR_INVALID_ARG


def test_keyed_step_and_reset_with_fused_status_match_oracle(lib):
    """env2048_step_state_i32_keyed_status / env2048_reset_keyed_i32_status: the plain results are unchanged and the
    observation / (N, 4) legal bytes of the boards the call leaves behind equal the oracle's status of those boards --
    frozen (done) games included; pinned HOST memory works as the key / action source (what the mirrors stage small
    inputs through)."""
    rng = np.random.default_rng(21)
    n = 9_001
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    values = oracle.unpack_values(random_boards(rng, n)).astype(np.int32)
    actions = rng.integers(0, 4, size=n).astype(np.int32)
    step_count = rng.integers(0, 500, size=n).astype(np.int32)
    done = (rng.random(n) < 0.3).astype(np.uint8)
    total = rng.integers(0, 5000, size=n).astype(np.float32)
    ref = oracle.step_state_i32_keyed(values, actions, step_count, done, total, keys)
    status = oracle.status_batch(oracle.pack_values(ref['boards']).reshape(n, 16))
    want_legal4 = ((status['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8)
    k_pinned = torch.from_numpy(keys.view(np.int32).copy()).pin_memory()  # device-accessible host memory
    a_pinned = torch.from_numpy(actions.copy()).pin_memory()
    b_in, sc_in, d_in, tr_in = cu(values), cu(step_count), cu(done), cu(total)
    b_out = torch.full((n, 16), -5, dtype=torch.int32, device='cuda')
    sc_out = torch.full((n,), -5, dtype=torch.int32, device='cuda')
    d_out = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    tr_out = torch.full((n,), -5.0, dtype=torch.float32, device='cuda')
    reward = torch.full((n,), -1.0, dtype=torch.float32, device='cuda')
    max_tile = torch.full((n,), -1, dtype=torch.int32, device='cuda')
    obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda')
    legal4 = torch.full((n, 4), 7, dtype=torch.uint8, device='cuda')
    lib.step_state_i32_keyed_status(DEV, stream(), n, p(k_pinned), p(b_in), p(b_out), p(a_pinned), p(sc_in), p(sc_out), p(d_in),
                                    p(d_out), p(tr_in), p(tr_out), p(reward), p(max_tile), p(obs), p(legal4))
    torch.cuda.synchronize()
    for got, key in ((b_out, 'boards'), (sc_out, 'step_count'), (d_out, 'done'), (tr_out, 'total_reward'), (reward, 'reward'),
                     (max_tile, 'max_tile')):
        np.testing.assert_array_equal(got.cpu().numpy(), ref[key], err_msg=key)
    np.testing.assert_array_equal(obs.cpu().numpy(), status['obs'])
    np.testing.assert_array_equal(legal4.cpu().numpy(), want_legal4)
    # obs only / legal only
    obs.fill_(-1.0)
    lib.step_state_i32_keyed_status(DEV, stream(), n, p(k_pinned), p(b_in), p(b_out), p(a_pinned), p(sc_in), p(sc_out), p(d_in),
                                    p(d_out), p(tr_in), p(tr_out), p(reward), p(max_tile), p(obs), None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(obs.cpu().numpy(), status['obs'])
    # reset + status
    fresh = torch.full((n, 16), -1, dtype=torch.int32, device='cuda')
    lib.reset_keyed_i32_status(DEV, stream(), n, p(k_pinned), p(fresh), p(obs), p(legal4))
    torch.cuda.synchronize()
    want = oracle.reset_keyed_i32(keys)
    np.testing.assert_array_equal(fresh.cpu().numpy(), want)
    st0 = oracle.status_batch(oracle.pack_values(want).reshape(n, 16))
    np.testing.assert_array_equal(obs.cpu().numpy(), st0['obs'])
    np.testing.assert_array_equal(legal4.cpu().numpy(), ((st0['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8))
    with pytest.raises(capi.Env2048Error) as e:
        lib.reset_keyed_i32_status(DEV, stream(), n, p(k_pinned), p(fresh), p(obs) + 4, None)
    assert e.value.code == capi.ERR_ALIGNMENT


def test_mirrored_step_and_reset_write_the_host_block_too(lib):
    """env2048_step_state_i32_keyed_mirrored / env2048_reset_state_i32_keyed: all outputs of a call live in ONE device block;
    with host_mirror_delta = (pinned host block) - (device block) every output is also written to the host block, which
    after one synchronisation equals the device block byte for byte -- and both equal the oracle (frozen games included).
    delta = 0: no mirror; a misaligned delta and the in-place form are refused."""
    rng = np.random.default_rng(77)
    for n in (4, 8, 333):
        keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
        values = oracle.unpack_values(random_boards(rng, n)).astype(np.int32)
        actions = rng.integers(0, 4, size=n).astype(np.int32)
        step_count = rng.integers(0, 500, size=n).astype(np.int32)
        done = (rng.random(n) < 0.3).astype(np.uint8)
        total = rng.integers(0, 5000, size=n).astype(np.float32)
        ref = oracle.step_state_i32_keyed(values, actions, step_count, done, total, keys)
        status = oracle.status_batch(oracle.pack_values(ref['boards']).reshape(n, 16))
        # one block: board | obs | step_count | total | done | max_tile | reward | legal4, each 16-byte aligned
        sizes = [('board', 64 * n), ('obs', 64 * n), ('step_count', 4 * n), ('total', 4 * n), ('done', n), ('max_tile', 4 * n), ('reward', 4 * n),
                 ('legal', 4 * n)]
        off, at = {}, 0
        for name, size in sizes:
            off[name] = at
            at = (at + size + 15) // 16 * 16
        block = torch.full((at,), 0xEE, dtype=torch.uint8, device='cuda')
        host = torch.full((at,), 0xDD, dtype=torch.uint8).pin_memory()
        delta = host.data_ptr() - block.data_ptr()
        assert delta % 16 == 0
        q = {name: block.data_ptr() + o for name, o in off.items()}
        k, a = cu(keys.view(np.int32)), cu(actions)
        b_in, sc_in, d_in, tr_in = cu(values), cu(step_count), cu(done), cu(total)
        lib.step_state_i32_keyed_mirrored(DEV, stream(), n, p(k), p(b_in), q['board'], p(a), p(sc_in), q['step_count'], p(d_in), q['done'],
                                          p(tr_in), q['total'], q['reward'], q['max_tile'], q['obs'], q['legal'], delta)
        torch.cuda.synchronize()
        dev_bytes, host_bytes = block.cpu().numpy(), host.numpy()

        def field(buf, name, dtype, shape):
            return buf[off[name]:off[name] + dict(sizes)[name]].view(dtype).reshape(shape)

        for buf, where in ((dev_bytes, 'device'), (host_bytes, 'host mirror')):
            np.testing.assert_array_equal(field(buf, 'board', np.int32, (n, 16)), ref['boards'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'step_count', np.int32, (n,)), ref['step_count'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'done', np.uint8, (n,)), ref['done'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'total', np.float32, (n,)), ref['total_reward'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'reward', np.float32, (n,)), ref['reward'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'max_tile', np.int32, (n,)), ref['max_tile'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'obs', np.float32, (n, 16)), status['obs'], err_msg=where)
            np.testing.assert_array_equal(field(buf, 'legal', np.uint8, (n, 4)), ((status['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8),
                                          err_msg=where)
        # bytes the call does not own (padding between the arrays) are untouched on both sides
        owned = np.zeros(at, dtype=bool)
        for name, size in sizes:
            owned[off[name]:off[name] + size] = True
        assert (dev_bytes[~owned] == 0xEE).all() and (host_bytes[~owned] == 0xDD).all()
        # reset: board + bookkeeping zeros + status, mirrored
        block.fill_(0xEE)
        host.fill_(0xDD)
        lib.reset_state_i32_keyed(DEV, stream(), n, p(k), q['board'], q['step_count'], q['done'], q['total'], q['obs'], q['legal'], delta)
        torch.cuda.synchronize()
        want = oracle.reset_keyed_i32(keys)
        st0 = oracle.status_batch(oracle.pack_values(want).reshape(n, 16))
        for buf, where in ((block.cpu().numpy(), 'device'), (host.numpy(), 'host mirror')):
            np.testing.assert_array_equal(field(buf, 'board', np.int32, (n, 16)), want, err_msg=where)
            assert not field(buf, 'step_count', np.int32, (n,)).any() and not field(buf, 'done', np.uint8, (n,)).any(), where
            assert not field(buf, 'total', np.float32, (n,)).any(), where
            np.testing.assert_array_equal(field(buf, 'obs', np.float32, (n, 16)), st0['obs'], err_msg=where)
        # no mirror: the host block stays as it was
        host.fill_(0xDD)
        lib.step_state_i32_keyed_mirrored(DEV, stream(), n, p(k), p(b_in), q['board'], p(a), p(sc_in), q['step_count'], p(d_in), q['done'],
                                          p(tr_in), q['total'], q['reward'], q['max_tile'], q['obs'], q['legal'], 0)
        torch.cuda.synchronize()
        assert (host.numpy() == 0xDD).all()
        np.testing.assert_array_equal(field(block.cpu().numpy(), 'board', np.int32, (n, 16)), ref['boards'])
    with pytest.raises(capi.Env2048Error) as e:
        lib.step_state_i32_keyed_mirrored(DEV, stream(), n, p(k), p(b_in), q['board'], p(a), p(sc_in), q['step_count'], p(d_in), q['done'],
                                          p(tr_in), q['total'], q['reward'], q['max_tile'], q['obs'], q['legal'], delta + 4)
    assert e.value.code == capi.ERR_ALIGNMENT
    with pytest.raises(capi.Env2048Error) as e:  # in place + mirror: refused
        lib.step_state_i32_keyed_mirrored(DEV, stream(), n, p(k), p(b_in), p(b_in), p(a), p(sc_in), p(sc_in), p(d_in), p(d_in),
                                          p(tr_in), p(tr_in), q['reward'], q['max_tile'], None, None, delta)
    assert e.value.code == capi.ERR_INVALID_ARG


def test_signalled_launches_publish_their_serial_after_the_mirror(lib):
    """env2048_*_signalled: a single-workgroup launch with a host mirror stores `sequence_value` into
    the caller's pinned completion word AFTER all its result stores -- so a host that sees the value WITHOUT synchronising the
    stream finds the complete results in the mirror.  200 back-to-back moves per size, each checked against the oracle from
    the mirror alone (no stream synchronisation anywhere in the loop); n = 1 ... 512 covers one to eight waves (the barrier
    in front of the word spans the whole workgroup, lanes beyond n included).  Refused: a word without a mirror, n > 512,
    a misaligned word.  env2048_wait_sequence times out (returns False) on a value nobody writes."""
    rng = np.random.default_rng(404)
    for n in (1, 2, 8, 63, 64, 65, 100, 333, 512):
        sizes = [('board', 64 * n), ('obs', 64 * n), ('step_count', 4 * n), ('total', 4 * n), ('done', n), ('max_tile', 4 * n), ('reward', 4 * n),
                 ('legal', 4 * n)]
        off, at = {}, 0
        for name, size in sizes:
            off[name] = at
            at = (at + size + 15) // 16 * 16
        blocks = [torch.zeros((at,), dtype=torch.uint8, device='cuda') for _ in range(2)]  # ping-pong: out-of-place steps
        hosts = [torch.zeros((at + 16,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        for b, h in zip(blocks, hosts):
            assert (h.data_ptr() - b.data_ptr()) % 16 == 0

        def field(buf, name, dtype, shape):
            return buf[off[name]:off[name] + dict(sizes)[name]].view(dtype).reshape(shape)

        keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
        k = cu(keys.view(np.int32))
        q = {name: blocks[0].data_ptr() + o for name, o in off.items()}
        serial = 1
        lib.reset_state_i32_keyed_signalled(DEV, stream(), n, p(k), q['board'], q['step_count'], q['done'], q['total'], q['obs'], q['legal'],
                                            hosts[0].data_ptr() - blocks[0].data_ptr(), hosts[0].data_ptr() + at, serial)
        assert lib.wait_sequence(hosts[0].data_ptr() + at, serial, 2_000_000), 'the reset never published its serial'
        host = hosts[0].numpy()
        state = {'boards': field(host, 'board', np.int32, (n, 16)).copy(), 'step_count': field(host, 'step_count', np.int32, (n,)).copy(),
                 'done': field(host, 'done', np.uint8, (n,)).copy(), 'total_reward': field(host, 'total', np.float32, (n,)).copy()}
        np.testing.assert_array_equal(state['boards'], oracle.reset_keyed_i32(keys))
        cur = 0
        for move in range(200 if n <= 100 else 40):
            keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
            actions = rng.integers(0, 4, size=n).astype(np.int32)
            k, a = cu(keys.view(np.int32)), cu(actions)
            src = {name: blocks[cur].data_ptr() + o for name, o in off.items()}
            dst = {name: blocks[1 - cur].data_ptr() + o for name, o in off.items()}
            serial += 1
            word = hosts[1 - cur].data_ptr() + at
            torch.cuda.current_stream().synchronize()  # (k / a uploaded; from here on NO synchronisation until the mirror is read)
            lib.step_state_i32_keyed_signalled(DEV, stream(), n, p(k), src['board'], dst['board'], p(a), src['step_count'], dst['step_count'],
                                               src['done'], dst['done'], src['total'], dst['total'], dst['reward'], dst['max_tile'], dst['obs'],
                                               dst['legal'], hosts[1 - cur].data_ptr() - blocks[1 - cur].data_ptr(), word, serial)
            assert lib.wait_sequence(word, serial, 2_000_000), (n, move)
            host = hosts[1 - cur].numpy()
            ref = oracle.step_state_i32_keyed(state['boards'], actions, state['step_count'], state['done'], state['total_reward'], keys)
            np.testing.assert_array_equal(field(host, 'board', np.int32, (n, 16)), ref['boards'], err_msg=f'n={n} move={move}')
            np.testing.assert_array_equal(field(host, 'reward', np.float32, (n,)), ref['reward'])
            np.testing.assert_array_equal(field(host, 'done', np.uint8, (n,)), ref['done'])
            np.testing.assert_array_equal(field(host, 'step_count', np.int32, (n,)), ref['step_count'])
            np.testing.assert_array_equal(field(host, 'total', np.float32, (n,)), ref['total_reward'])
            np.testing.assert_array_equal(field(host, 'max_tile', np.int32, (n,)), ref['max_tile'])
            status = oracle.status_batch(oracle.pack_values(ref['boards']).reshape(n, 16))
            np.testing.assert_array_equal(field(host, 'obs', np.float32, (n, 16)), status['obs'])
            state = {'boards': ref['boards'], 'step_count': ref['step_count'], 'done': ref['done'], 'total_reward': ref['total_reward']}
            cur = 1 - cur
        torch.cuda.synchronize()
        np.testing.assert_array_equal(field(blocks[cur].cpu().numpy(), 'board', np.int32, (n, 16)), state['boards'])  # device copy too
    assert lib.wait_sequence(word, serial + 1000, 200) is False  # nobody writes that value: a timeout, not a hang
    n = 8
    args = (DEV, stream(), n, p(k[:n].contiguous()), src['board'], dst['board'], p(a[:n].contiguous()), src['step_count'], dst['step_count'],
            src['done'], dst['done'], src['total'], dst['total'], dst['reward'], dst['max_tile'], dst['obs'], dst['legal'])
    with pytest.raises(capi.Env2048Error, match='goes with a host mirror'):
        lib.step_state_i32_keyed_signalled(*args, 0, word, 1)
    with pytest.raises(capi.Env2048Error, match='8-byte aligned'):
        lib.step_state_i32_keyed_signalled(*args, hosts[0].data_ptr() - blocks[0].data_ptr(), word + 4, 1)
    big_keys = cu(np.zeros((513, 2), dtype=np.int32))
    big = torch.zeros((513 * 200,), dtype=torch.uint8, device='cuda')
    with pytest.raises(capi.Env2048Error, match='single-workgroup'):
        lib.reset_state_i32_keyed_signalled(DEV, stream(), 513, p(big_keys), big.data_ptr(), None, None, None, None, None, 16, word, 1)
    torch.cuda.synchronize()


def test_step_state_i32_matches_oracle(lib):
    """Fused reinforce/game step on int32 value boards: done-guard, bookkeeping, JAX is_done."""
    rng = np.random.default_rng(11)
    n = 30_001
    exps = random_boards(rng, n)
    values = oracle.unpack_values(exps).astype(np.int32)
    actions = rng.integers(-3, 9, size=n).astype(np.int32)  # any int: taken mod 4 (two's complement & 3)
    step_count = rng.integers(0, 500, size=n).astype(np.int32)
    done = (rng.random(n) < 0.2).astype(np.uint8)
    total = rng.integers(0, 5000, size=n).astype(np.float32)
    seed, step, env_base = 0xABCDEF, 77, 13
    ref = oracle.step_state_i32(values, actions, step_count, done, total, seed, step, env_base=env_base)
    b, a = cu(values), cu(actions)
    sc, d, tr = cu(step_count), cu(done), cu(total)
    reward = torch.full((n,), -1.0, dtype=torch.float32, device='cuda')
    max_tile = torch.full((n,), -1, dtype=torch.int32, device='cuda')
    lib.step_state_i32(DEV, stream(), n, env_base, p(b), p(a), p(sc), p(d), p(tr), p(reward), p(max_tile), seed, step)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(b.cpu().numpy(), ref['boards'])
    np.testing.assert_array_equal(sc.cpu().numpy(), ref['step_count'])
    np.testing.assert_array_equal(d.cpu().numpy(), ref['done'])
    np.testing.assert_array_equal(tr.cpu().numpy(), ref['total_reward'])
    np.testing.assert_array_equal(reward.cpu().numpy(), ref['reward'])
    np.testing.assert_array_equal(max_tile.cpu().numpy(), ref['max_tile'])
    frozen = done.astype(bool)
    np.testing.assert_array_equal(ref['boards'][frozen], values[frozen])  # the guard really froze them
    assert not ref['reward'][frozen].any()
    # the functional form: inputs only read, every output written (poisoned first), same results
    b_in, sc_in, d_in, tr_in = cu(values), cu(step_count), cu(done), cu(total)
    b_out = torch.full((n, 16), -5, dtype=torch.int32, device='cuda')
    sc_out = torch.full((n,), -5, dtype=torch.int32, device='cuda')
    d_out = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    tr_out = torch.full((n,), -5.0, dtype=torch.float32, device='cuda')
    reward.fill_(-1.0)
    max_tile.fill_(-1)
    lib.step_state_i32_to(DEV, stream(), n, env_base, p(b_in), p(b_out), p(a), p(sc_in), p(sc_out), p(d_in), p(d_out), p(tr_in),
                          p(tr_out), p(reward), p(max_tile), seed, step)
    torch.cuda.synchronize()
    for got, key in ((b_out, 'boards'), (sc_out, 'step_count'), (d_out, 'done'), (tr_out, 'total_reward'), (reward, 'reward'),
                     (max_tile, 'max_tile')):
        np.testing.assert_array_equal(got.cpu().numpy(), ref[key], err_msg='functional ' + key)
    for got, want in ((b_in, values), (sc_in, step_count), (d_in, done), (tr_in, total)):
        np.testing.assert_array_equal(got.cpu().numpy().reshape(want.shape), want)  # inputs untouched
    # optional arrays left out: both of a pair or neither
    lib.step_state_i32_to(DEV, stream(), n, env_base, p(b_in), p(b_out), p(a), None, None, p(d_in), p(d_out), None, None,
                          p(reward), None, seed, step)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(b_out.cpu().numpy(), ref['boards'])
    with pytest.raises(capi.Env2048Error):
        lib.step_state_i32_to(DEV, stream(), n, env_base, p(b_in), p(b_out), p(a), p(sc_in), None, p(d_in), p(d_out), None, None,
                              p(reward), None, seed, step)


def test_status_i32_matches_oracle(lib):
    """obs / (n,4) legal bytes / JAX done on int32 value boards, every output optional."""
    rng = np.random.default_rng(12)
    n = 20_011
    exps = random_boards(rng, n)
    values = oracle.unpack_values(exps).astype(np.int32)
    ref = oracle.status_batch(exps)
    want_legal = ((ref['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8)
    b = cu(values)
    obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda')
    legal = torch.full((n, 4), 9, dtype=torch.uint8, device='cuda')
    done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    lib.status_i32(DEV, stream(), n, p(b), p(obs), p(legal), p(done))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(obs.cpu().numpy(), ref['obs'])
    np.testing.assert_array_equal(legal.cpu().numpy(), want_legal)
    np.testing.assert_array_equal(done.cpu().numpy(), (ref['legal'] == 0).astype(np.uint8))
    assert bool(legal.view(torch.bool).cpu().numpy()[5].tolist() == want_legal[5].astype(bool).tolist())
    legal.fill_(9)
    lib.status_i32(DEV, stream(), n, p(b), None, p(legal), None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(legal.cpu().numpy(), want_legal)


@pytest.mark.parametrize('n,env_base,shift', [(1, 0, 0), (3, 1, 0), (4, 4, 0), (5, 7, 1), (64, 2, 3), (257, (1 << 33) + 3, 0), (1000, 5, 2), (4096, 0, 0), (4099, 13, 1)])
def test_random_actions_are_the_low_bits_of_the_step_words(lib, n, env_base, shift):
    """env2048_random_actions (one lane per step block, dword stores where the batch and the destination allow) against the
    oracle's step words, for batches that start and end inside a block and destinations at any byte offset; no byte outside
    the batch is written."""
    seed, step = 0xABCDEF0123456789, (1 << 34) + 17
    arena = torch.full((n + shift + 8,), 0x77, dtype=torch.uint8, device='cuda')
    lib.random_actions(DEV, stream(), n, env_base, seed, step, arena[shift:].data_ptr())
    host = arena.cpu().numpy()
    want = np.array([oracle.step_word(seed, env_base + i, step) & 3 for i in range(n)], dtype=np.uint8)
    np.testing.assert_array_equal(host[shift:shift + n], want)
    assert (host[:shift] == 0x77).all() and (host[shift + n:] == 0x77).all()


def test_afterstate_i32_golden(lib, golden):
    """latent_state / slide_and_merge of the JAX-shaped API on int32 value boards (env2048_afterstate_i32) against the
    reference's own afterstates: one move for the whole batch, one move per board, in place, outputs optional."""
    g = golden['afterstates']
    exps = g['boards']
    n = exps.shape[0]
    values = cu(oracle.unpack_values(exps).astype(np.int32))
    want = [oracle.unpack_values(g['after'][:, a]).astype(np.int32) for a in range(4)]
    out = torch.full((n, 16), -7, dtype=torch.int32, device='cuda')
    reward = torch.full((n,), -7, dtype=torch.int32, device='cuda')
    changed = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    for a in range(4):
        lib.afterstate_i32(DEV, stream(), n, p(values), None, a + 4 * (a & 1), p(out), p(reward), p(changed))  # (only the low two bits count)
        np.testing.assert_array_equal(out.cpu().numpy(), want[a])
        np.testing.assert_array_equal(reward.cpu().numpy(), g['reward'][:, a])
        np.testing.assert_array_equal(changed.cpu().numpy(), g['valid'][:, a])
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 4, size=n).astype(np.int32)
    inplace = values.clone()
    per_board = cu(acts)
    lib.afterstate_i32(DEV, stream(), n, p(inplace), p(per_board), 0, p(inplace), None, None)
    mixed = np.stack(want, axis=1)[np.arange(n), acts]
    np.testing.assert_array_equal(inplace.cpu().numpy(), mixed)
    lib.afterstate_i32(DEV, stream(), 0, None, None, 0, None, None, None)  # n == 0: nothing
    with pytest.raises(capi.Env2048Error):
        lib.afterstate_i32(DEV, stream(), n, p(values), None, 0, None, None, None)
    with pytest.raises(capi.Env2048Error, match='aligned'):
        lib.afterstate_i32(DEV, stream(), n - 1, p(values) + 4, None, 0, p(out), None, None)


def test_config4_slicing_at_8x2pow20(lib):
    """BASELINE config 4 on one GPU: 8 388 608 boards stepped as 8 contiguous slices (what 8 ranks would
    do, each with env_base = its slice start) equal the unsliced batch -- boards, rewards and done flags."""
    n, G, seed, T = 8 << 20, 8, 0x2048, 6
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    whole = torch.empty((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(DEV, stream(), n, 0, p(whole), None, seed, 0)
    lib.rollout(DEV, stream(), n, 0, p(whole), seed, 0, 90, None, None, None, flags)  # mid-game
    sliced = whole.clone()
    rw = torch.zeros((n,), dtype=torch.int32, device='cuda')
    dw = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    rs, ds = torch.zeros_like(rw), torch.zeros_like(dw)
    per = n // G
    for t in range(90, 90 + T):
        lib.step(DEV, stream(), n, 0, p(whole), None, p(rw), p(dw), None, None, seed, t, flags)
        for g in range(G):
            lo = g * per
            lib.step(DEV, stream(), per, lo, p(sliced) + 16 * lo, None, p(rs) + 4 * lo, p(ds) + lo, None, None, seed, t, flags)
        assert torch.equal(whole, sliced) and torch.equal(rw, rs) and torch.equal(dw, ds), f'step {t}'
    assert int(dw.sum()) > 0  # some games ended in the last step: the reset path was exercised at this size


def test_soak_rollout_3000_steps(lib):
    """8192 games x 3000 steps (about 20 finished games each): the fused rollout against the oracle's
    rollout, compared every 100 steps -- boards, return sums, episode and valid-move counts."""
    n, seed, chunk = 8192, 424242, 100
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    ref, _ = oracle.reset_batch(n, seed, 0, threads=16)
    b = cu(ref)
    ret = torch.zeros((n,), dtype=torch.int64, device='cuda')
    eps = torch.zeros((n,), dtype=torch.int32, device='cuda')
    moves = torch.zeros((n,), dtype=torch.int32, device='cuda')
    total_eps = 0
    for step0 in range(0, 3000, chunk):
        lib.rollout(DEV, stream(), n, 0, p(b), seed, step0, chunk, p(ret), p(eps), p(moves), flags)
        r = oracle.rollout_batch(ref, seed, step0, chunk, flags=flags, threads=16)
        ref = r['boards']
        torch.cuda.synchronize()
        assert np.array_equal(b.cpu().numpy(), ref), f'boards differ after step {step0 + chunk}'
        assert np.array_equal(ret.cpu().numpy(), r['return_sum'])
        assert np.array_equal(eps.cpu().numpy().astype(np.uint32), r['episodes'])
        assert np.array_equal(moves.cpu().numpy().astype(np.uint32), r['valid_moves'])
        total_eps += int(r['episodes'].sum())
    assert total_eps > 15 * n
