"""
Pin the CPU oracle (oracle/env2048_oracle.c) against

* the golden fixtures generated from the imported reference NumPy engine (tests/golden/*.npz,
  made by oracle/gen_golden.py), and
* the known-answer vectors of the reference's own tests, transcribed as data.

No GPU needed.  If these fail, nothing else in the suite means anything.
"""

import hashlib

import numpy as np
import pytest

import oracle
from conftest import GOLDEN

V = oracle.pack_values  # tile values -> exponents
U = oracle.unpack_values


def test_fixture_checksums():
    for line in (GOLDEN / 'SHA256SUMS').read_text().splitlines():
        digest, name = line.split()
        assert hashlib.sha256((GOLDEN / name).read_bytes()).hexdigest() == digest, name


# ---------------------------------------------------------------------------- Philox contract
@pytest.mark.parametrize(
    'ctr,key,expect',
    [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ],
)
def test_philox_random123_kat(ctr, key, expect):
    """Random123 known-answer vectors for Philox4x32-10."""
    assert oracle.philox4x32_10(ctr, key).tolist() == expect


def test_env_block_counter_layout():
    seed, index, step = 0x0123456789ABCDEF, (3 << 32) | 17, (5 << 32) | 9
    for stream in (0, 1):
        direct = oracle.philox4x32_10([17, 9, 5, stream + 2 * 3], [0x89ABCDEF, 0x01234567])
        assert oracle.env_block(seed, index, step, stream).tolist() == direct.tolist()


def test_step_word_sharing():
    """Environment e draws word (e & 3) of the stream-0 block with index e >> 2."""
    seed, step = 0xFEEDFACE12345678, (1 << 35) + 4
    for env in [0, 1, 2, 3, 4, 1001, (7 << 32) + 6, (1 << 34) + 3]:
        blk = oracle.env_block(seed, env >> 2, step, 0)
        assert oracle.step_word(seed, env, step) == int(blk[env & 3])


def test_spawn_from_word():
    """k = hi32((w & ~3) * n), tile 4 iff lo32 < floor(0.1 * 2^32), action = w & 3."""
    T = oracle.SPAWN4_THRESHOLD
    assert oracle.spawn_from_word(0, 16) == (0, 2, 0)  # lo32 = 0 < T
    assert oracle.spawn_from_word(0xFFFFFFFF, 16) == (15, 1, 3)
    assert oracle.spawn_from_word(0x80000002, 3) == (1, 1, 2)  # product 0x1_8000_0000: lo32 = 2^31 >= T
    for w, n in [(0x12345679, 7), (0xDEADBEEF, 1), (0x0000FFFD, 15), (0xC0000000, 4)]:
        prod = (w & ~3) * n
        assert oracle.spawn_from_word(w, n) == (prod >> 32, 2 if (prod & 0xFFFFFFFF) < T else 1, w & 3)
    # distribution over one million consecutive counters: uniform cells, P(4) = 0.1
    rng = np.random.default_rng(3)
    ws = rng.integers(0, 2**32, size=200_000, dtype=np.uint64)
    prod = (ws & ~np.uint64(3)) * np.uint64(7)
    k = (prod >> np.uint64(32)).astype(np.int64)
    four = (prod & np.uint64(0xFFFFFFFF)) < T
    assert np.bincount(k, minlength=7).min() > 0.95 * 200_000 / 7
    assert 0.097 < four.mean() < 0.103
    assert abs(np.corrcoef(k, four)[0, 1]) < 0.01 and abs(np.corrcoef(ws & np.uint64(3), four)[0, 1]) < 0.01


def test_spawn_from_words():
    assert oracle.spawn_from_words(0, 0, 16) == (0, 2)
    assert oracle.spawn_from_words(0xFFFFFFFF, 0xFFFFFFFF, 16) == (15, 1)
    assert oracle.spawn_from_words(0x80000000, oracle.SPAWN4_THRESHOLD - 1, 3) == (1, 2)
    assert oracle.spawn_from_words(0x80000000, oracle.SPAWN4_THRESHOLD, 3) == (1, 1)


# ---------------------------------------------------------------------------- G1 rows
def test_rows_exhaustive(golden):
    g = golden['rows']
    base = int(g['max_e']) + 1
    merged, score, can_move = g['merged'], g['score'], g['can_move']
    assert merged.shape[0] == base**4
    idx = 0
    for a in range(base):
        for b in range(base):
            for c in range(base):
                for d in range(base):
                    s, out = oracle.merge_column([a, b, c, d])
                    padded = np.zeros(4, dtype=np.uint8)
                    padded[: len(out)] = out
                    assert s == score[idx] and padded.tolist() == merged[idx].tolist(), (a, b, c, d)
                    # can_move == "a left move changes the row" 
                    assert (padded.tolist() != [a, b, c, d]) == bool(can_move[idx])
                    idx += 1


# ---------------------------------------------------------------------------- G2 afterstates
def test_afterstates(golden):
    g = golden['afterstates']
    boards = g['boards']
    n = boards.shape[0]
    for a in range(4):
        res = oracle.afterstate_batch(boards, np.full(n, a, dtype=np.uint8))
        np.testing.assert_array_equal(res['boards'], g['after'][:, a])
        np.testing.assert_array_equal(res['reward'], g['reward'][:, a])
        np.testing.assert_array_equal(res['changed'], g['valid'][:, a])
    st = oracle.status_batch(boards)
    np.testing.assert_array_equal(st['legal'], g['legal'])
    np.testing.assert_array_equal(st['done'], g['done'])
    # legal mask bit a <=> move a valid (both straight from the reference)
    np.testing.assert_array_equal((g['legal'][:, None] >> np.arange(4)) & 1, g['valid'])
    for i in range(0, n, 7):
        for a in range(4):
            assert oracle.can_move(oracle.rot90(boards[i], a)) == bool(g['valid'][i, a])


# ---------------------------------------------------------------------------- G3 real next_state
def test_next_state_with_injected_spawn(golden):
    g = golden['next_state']
    boards, actions = g['boards'], g['actions']
    after = oracle.afterstate_batch(boards, actions)
    moved = g['spawn_k'] != 255
    np.testing.assert_array_equal(after['changed'].astype(bool), moved)
    k = np.where(moved, g['spawn_k'], 200).astype(np.uint8)  # 200: out of range -> no spawn
    out = oracle.spawn_batch(after['boards'], k, g['spawn_e'])
    np.testing.assert_array_equal(out['boards'], g['out'])
    np.testing.assert_array_equal(np.where(moved, after['reward'], 0), g['reward'])
    np.testing.assert_array_equal(oracle.status_batch(out['boards'])['done'], g['done'])
    assert set(np.unique(g['spawn_e'][moved]).tolist()) <= {1, 2}


# ---------------------------------------------------------------------------- G4 trajectories
@pytest.mark.parametrize('name', ['auto', 'plain'])
def test_trajectories(golden, name):
    g = golden['trajectories']
    seed, env_base, step0 = int(g[f'{name}_seed']), int(g[f'{name}_env_base']), int(g[f'{name}_step0'])
    flags = oracle.FLAG_AUTO_RESET if int(g[f'{name}_auto_reset']) else 0
    T, n = g[f'{name}_actions'].shape
    boards, _ = oracle.reset_batch(n, seed, step0, env_base=env_base)
    np.testing.assert_array_equal(boards, g[f'{name}_boards0'])
    total = np.zeros(n, dtype=np.int64)
    for t in range(T):
        res = oracle.step_batch(boards, g[f'{name}_actions'][t], seed, step0 + t, env_base=env_base, flags=flags,
                                want_legal=True, want_obs=True)
        boards = res['boards']
        np.testing.assert_array_equal(boards, g[f'{name}_boards'][t], err_msg=f't={t}')
        np.testing.assert_array_equal(res['reward'], g[f'{name}_reward'][t])
        np.testing.assert_array_equal(res['done'], g[f'{name}_done'][t])
        np.testing.assert_array_equal(res['legal'], g[f'{name}_legal'][t])
        np.testing.assert_array_equal(res['obs'], boards.astype(np.float32) / 16.0)
        total += res['reward']
        # on-device actions reproduce the recorded ones
        res2 = oracle.step_batch(g[f'{name}_boards'][t - 1] if t else g[f'{name}_boards0'], None, seed, step0 + t,
                                 env_base=env_base, flags=flags | oracle.FLAG_RANDOM_ACTIONS)
        np.testing.assert_array_equal(res2['boards'], boards)
    # the fused rollout agrees with T single steps
    ro = oracle.rollout_batch(g[f'{name}_boards0'], seed, step0, T, env_base=env_base, flags=flags)
    np.testing.assert_array_equal(ro['boards'], boards)
    np.testing.assert_array_equal(ro['return_sum'], total)
    np.testing.assert_array_equal(ro['episodes'], g[f'{name}_done'].sum(axis=0))


def test_resets(golden):
    g = golden['resets']
    boards, done = oracle.reset_batch(g['boards'].shape[0], int(g['seed']), int(g['step']), env_base=int(g['env_base']))
    np.testing.assert_array_equal(boards, g['boards'])
    assert not done.any()
    assert np.all((boards != 0).sum(axis=1) == 2)
    assert set(np.unique(boards).tolist()) <= {0, 1, 2}


def test_chance_outcomes_vs_reference_after_state(golden):
    """The reference's after_state itself (gameboard.py:132-171, called by oracle/gen_golden.py): enumeration order
    (empty cells row-major, value 2 then 4), outcome boards, float64 probabilities P(value)/n_empty, the full-board
    case.  The oracle (and the kernel it checks) emits float32 probabilities: (float)0.9 or 0.1 divided by n in
    float32 -- tolerance 2 ulp of float32 (2.4e-7 relative) against the reference's float64 value."""
    g = golden['chance_outcomes']
    res = oracle.chance_outcomes(g['boards'])
    np.testing.assert_array_equal(res['count'], g['count'])
    np.testing.assert_array_equal(res['boards'], g['outcomes'])
    np.testing.assert_allclose(res['prob'].astype(np.float64), g['prob'], rtol=2.4e-7, atol=0.0)
    assert int(g['count'].max()) == 32 and int((g['count'] == 1).sum()) > 50  # 16 empties down to none at all
    np.testing.assert_allclose(g['prob'].sum(axis=1), 1.0, atol=1e-12)


def test_reference_shaped_numpy_baseline_against_the_fixtures(golden):
    """oracle/numpy_shaped.py (the per-board NumPy engine bench.py times as the reference-shaped CPU baseline) against the
    reference-generated fixtures: afterstates + scores + validity, terminal flags, and full Philox-driven trajectories."""
    from oracle import numpy_shaped as ns

    g = golden['afterstates']
    rng = np.random.default_rng(0)
    for i in rng.choice(g['boards'].shape[0], size=400, replace=False):
        board = oracle.unpack_values(g['boards'][i]).reshape(4, 4)
        assert ns.terminal(board) == bool(g['done'][i])
        for a in range(4):
            after, gained = ns.afterstate(board, a)
            np.testing.assert_array_equal(oracle.pack_values(after).reshape(16), g['after'][i, a])
            assert gained == int(g['reward'][i, a])
            assert ns.left_move_possible(np.rot90(board, k=a)) == bool(g['valid'][i, a])
    t = golden['trajectories']
    seed, base, step0 = int(t['plain_seed']), int(t['plain_env_base']), int(t['plain_step0'])
    for j in (0, 7, 15):
        board = oracle.unpack_values(t['plain_boards0'][j]).reshape(4, 4)
        for step in range(t['plain_boards'].shape[0]):
            w = oracle.step_word(seed, base + j, step0 + step)
            board, reward, done = ns.transition(board, w & 3, w)
            assert np.array_equal(oracle.pack_values(board).reshape(16), t['plain_boards'][step, j])
            assert reward == int(t['plain_reward'][step, j]) and done == bool(t['plain_done'][step, j])


def test_batch_is_independent_of_slicing(golden):
    """Multi-GPU determinism: slices with the right env_base equal the whole."""
    g = golden['trajectories']
    b0 = g['auto_boards0']
    n = b0.shape[0]
    acts = g['auto_actions'][0]
    whole = oracle.step_batch(b0, acts, 99, 5, env_base=1000, flags=1)
    for lo, hi in [(0, n // 2), (n // 2, n)]:
        part = oracle.step_batch(b0[lo:hi], acts[lo:hi], 99, 5, env_base=1000 + lo, flags=1)
        np.testing.assert_array_equal(part['boards'], whole['boards'][lo:hi])
        np.testing.assert_array_equal(part['reward'], whole['reward'][lo:hi])


# ---------------------------------------------------------------------------- reference-test KATs
ROW_KATS = [  # tests/test_jax_game.py:35-82, tests/test_board.py:63-68, tests/test_environment.py:187-210
    ([0, 0, 0, 0], [0, 0, 0, 0], 0),
    ([0, 0, 2, 0], [2, 0, 0, 0], 0),
    ([2, 2, 0, 0], [4, 0, 0, 0], 4),
    ([2, 2, 2, 2], [4, 4, 0, 0], 8),
    ([2, 4, 8, 16], [2, 4, 8, 16], 0),
    ([2, 0, 2, 4], [4, 4, 0, 0], 4),
    ([4, 2, 2, 0], [4, 4, 0, 0], 4),
    ([2, 2, 4, 4], [4, 8, 0, 0], 12),
    ([0, 4, 0, 0], [4, 0, 0, 0], 0),
]


@pytest.mark.parametrize('row,expect,score', ROW_KATS)
def test_row_kats(row, expect, score):
    s, out = oracle.merge_column(V(row))
    padded = np.zeros(4, dtype=np.uint8)
    padded[: len(out)] = out
    assert s == score and U(padded).tolist() == expect


def test_merge_column_is_unpadded():
    """tests/test_board.py:63-68: merge_column([2,2,4,4]) -> 12, [4, 8] (length 2)."""
    s, out = oracle.merge_column(V([2, 2, 4, 4]))
    assert s == 12 and U(out).tolist() == [4, 8]
    s, out = oracle.merge_column(V([0, 4, 0, 0]))  # tests/test_environment.py:195-201
    assert s == 0 and U(out).tolist() == [4]
    s, out = oracle.merge_column(V([0, 0, 0, 0]))  # tests/test_environment.py:187-193
    assert s == 0 and len(out) == 0


BOARD_KATS = [
    # tests/test_board.py:70-76
    ([[2, 2, 4, 4], [0, 2, 2, 4], [2, 0, 0, 2], [2, 2, 2, 2]], [[4, 8, 0, 0], [4, 4, 0, 0], [4, 0, 0, 0], [4, 4, 0, 0]], 28),
    # tests/test_jax_game.py:88-95
    ([[2, 2, 4, 4], [4, 4, 2, 2], [2, 4, 2, 4], [4, 2, 4, 2]], [[4, 8, 0, 0], [8, 4, 0, 0], [2, 4, 2, 4], [4, 2, 4, 2]], 24),
    # tests/test_environment.py:212-219
    ([[2, 2, 0, 0], [4, 4, 0, 0], [8, 8, 0, 0], [16, 16, 0, 0]], [[4, 0, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0], [32, 0, 0, 0]], 60),
]


@pytest.mark.parametrize('board,expect,score', BOARD_KATS)
def test_slide_and_merge_kats(board, expect, score):
    s, out = oracle.slide_and_merge(V(board).reshape(16))
    assert s == score and U(out).reshape(4, 4).tolist() == expect


DIRECTION_KATS = [  # tests/test_jax_game.py:101-135 -- the authoritative action -> direction pin
    (0, [[0, 2, 0, 2], [0] * 4, [0] * 4, [0] * 4], [[4, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4]),
    (1, [[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4], [[4, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4]),
    (2, [[2, 0, 0, 2], [0] * 4, [0] * 4, [0] * 4], [[0, 0, 0, 4], [0] * 4, [0] * 4, [0] * 4]),
    (3, [[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4], [[0] * 4, [0] * 4, [0] * 4, [4, 0, 0, 0]]),
]


@pytest.mark.parametrize('action,board,expect', DIRECTION_KATS)
def test_direction_kats(action, board, expect):
    out, reward = oracle.latent_state(V(board).reshape(16), action)
    assert U(out).reshape(4, 4).tolist() == expect and reward == 4


def test_legality_kats():
    b = V([[2, 0, 0, 0], [2, 0, 0, 0], [0] * 4, [0] * 4]).reshape(16)  # tests/test_move.py:13-23
    assert oracle.legal_actions_mask(b) == 0b1110
    b = V([[2, 0, 0, 0], [4, 0, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0]]).reshape(16)  # tests/test_environment.py:121-136,175-181
    assert oracle.legal_actions_mask(b) & 1 == 0
    checker = V([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 2]]).reshape(16)  # tests/test_jax_game.py:147-151
    assert oracle.legal_actions_mask(checker) == 0 and oracle.is_done(checker)
    almost = V([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 2, 2]]).reshape(16)  # tests/test_jax_game.py:174-177
    assert not oracle.is_done(almost)
    # all-zero board: every move illegal (tests/test_jax_game.py:141-145); NOT done in the NumPy engine (App. A.5)
    assert oracle.legal_actions_mask(np.zeros(16, dtype=np.uint8)) == 0
    assert not oracle.is_done(np.zeros(16, dtype=np.uint8))


def test_terminal_kats():
    stair = V(np.array([2 ** (i + 1) for i in range(16)]).reshape(4, 4)).reshape(16)  # tests/test_board.py:53-56
    assert oracle.is_done(stair)
    b = V([[2, 0, 0, 0], [2, 0, 0, 0], [0] * 4, [0] * 4]).reshape(16)  # tests/test_board.py:58-61
    assert not oracle.is_done(b)
    one_empty = stair.copy()
    one_empty[5] = 0  # tests/test_environment.py:231-236
    assert not oracle.is_done(one_empty)
    pair = V([[2, 2, 4, 8], [16, 32, 64, 128], [256, 512, 1024, 2048], [4096, 8192, 16384, 32768]]).reshape(16)
    assert not oracle.is_done(pair)  # tests/test_environment.py:238-243


def test_step_kats():
    # tests/test_jax_game.py:212-225: up on two stacked 2s -> [0,0]==4, exactly 2 tiles, reward 4
    b = V([[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4]).reshape(1, 16)
    res = oracle.step_batch(b, [1], 11, 0)
    assert res['boards'][0, 0] == 2 and (res['boards'] != 0).sum() == 2 and res['reward'][0] == 4
    # tests/test_environment.py:138-149: [2,2,0,0] left -> reward == 4
    b = V([[2, 2, 0, 0], [0] * 4, [0] * 4, [0] * 4]).reshape(1, 16)
    res = oracle.step_batch(b, [0], 42, 0)
    assert res['reward'][0] == 4 and res['boards'][0, 0] == 2
    # tests/test_environment.py:121-136 / tests/test_board.py:45-51: invalid move -> unchanged, 0, no spawn
    b = V([[2, 0, 0, 0], [4, 0, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0]]).reshape(1, 16)
    res = oracle.step_batch(b, [0], 42, 0)
    np.testing.assert_array_equal(res['boards'], b)
    assert res['reward'][0] == 0 and res['done'][0] == 0
    # action is taken mod 4 like rot90
    r4 = oracle.step_batch(b, [4 + 1], 42, 0)
    r1 = oracle.step_batch(b, [1], 42, 0)
    np.testing.assert_array_equal(r4['boards'], r1['boards'])


def test_done_board_step_is_noop():
    """reinforce/game/env.py:103-104 done-guard == natural no-op of the NumPy engine."""
    checker = V([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 2]]).reshape(1, 16)
    for a in range(4):
        res = oracle.step_batch(checker, [a], 5, 9)
        np.testing.assert_array_equal(res['boards'], checker)
        assert res['reward'][0] == 0 and res['done'][0] == 1
    res = oracle.step_batch(checker, [0], 5, 9, flags=oracle.FLAG_AUTO_RESET)
    np.testing.assert_array_equal(res['boards'][0], oracle.reset_board(5, 0, 9))
    assert res['done'][0] == 1 and res['reward'][0] == 0


def test_obs_kats():
    """tests/test_jax_game.py:282-291: [2,4,8,16,0...] -> 1/16, 2/16, 3/16, 4/16, 0..."""
    b = V([[2, 4, 8, 16], [0] * 4, [0] * 4, [0] * 4]).reshape(1, 16)
    obs = oracle.status_batch(b)['obs'][0]
    assert obs[:4].tolist() == [1 / 16, 2 / 16, 3 / 16, 4 / 16] and not obs[4:].any()


def test_spawn_distribution():
    """tests/test_environment.py:249-279: spawns are 2 or 4 with 90/10 +-5 %; cells uniform."""
    boards, _ = oracle.reset_batch(20000, 1234, 0)
    vals = boards[boards != 0]
    frac4 = float((vals == 2).mean())
    assert 0.09 < frac4 < 0.11
    counts = (boards != 0).sum(axis=0)
    assert counts.min() > 0.9 * 2500 and counts.max() < 1.1 * 2500
