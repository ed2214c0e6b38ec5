"""
The ``twentyfortyeight`` drop-in surface (simulate_2048_amd.twentyfortyeight.*), exercised the way
the reference's tests/test_board.py, tests/test_environment.py and tests/test_move.py exercise the
original: same calls, same expected values (cited per test).  Each test runs with the oracle-backed
stand-in (host logic, CPU) and, on the GPU box, with the real HIP library.
"""

import numpy as np

import oracle
import pytest

from simulate_2048_amd.twentyfortyeight.core.gameboard import (
    TILE_SPAWN_PROBS,
    after_state,
    after_state_lazy,
    fill_cells,
    generate_outcome,
    is_done,
    latent_state,
    merge_column,
    next_state,
    slide_and_merge,
)
from simulate_2048_amd.twentyfortyeight.core.gamemove import can_move, illegal_actions, legal_actions, legal_actions_mask
from simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight import TwentyFortyEight
from simulate_2048_amd.twentyfortyeight.utils.binary import encode_flatten
from simulate_2048_amd.twentyfortyeight.utils.normalize import normalize_reward

pytestmark = pytest.mark.usefixtures('api_runtime')

LEFT_COLUMN = [[2, 0, 0, 0], [4, 0, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0]]
STAIRCASE = [[2, 4, 8, 16], [32, 64, 128, 256], [512, 1024, 2048, 4096], [8192, 16384, 32768, 65536]]


# ------------------------------------------------------------------ engine functions
def test_merge_column_kats():
    score, result = merge_column(np.array([2, 2, 4, 4]))  # tests/test_board.py:63-68
    assert score == 12
    np.testing.assert_array_equal(result, [4, 8])
    score, result = merge_column(np.array([0, 0, 0, 0]))  # tests/test_environment.py:187-193
    assert score == 0 and len(result) == 0
    score, result = merge_column(np.array([0, 4, 0, 0]))  # :195-201
    assert score == 0
    np.testing.assert_array_equal(result, [4])
    score, result = merge_column(np.array([2, 2, 2, 2]))  # :203-210
    assert score == 8
    np.testing.assert_array_equal(result, [4, 4])
    assert result.dtype == np.array([2]).dtype


def test_slide_and_merge_kats():
    board = np.array([[2, 2, 4, 4], [0, 2, 2, 4], [2, 0, 0, 2], [2, 2, 2, 2]])  # tests/test_board.py:70-76
    score, result = slide_and_merge(board)
    assert score == 28 and isinstance(score, float)
    np.testing.assert_array_equal(result, [[4, 8, 0, 0], [4, 4, 0, 0], [4, 0, 0, 0], [4, 4, 0, 0]])
    score, _ = slide_and_merge(np.array([[2, 2, 0, 0], [4, 4, 0, 0], [8, 8, 0, 0], [16, 16, 0, 0]]))
    assert score == 60  # tests/test_environment.py:212-219


def test_latent_state_directions():
    board = np.array([[2, 0, 0, 4], [0, 8, 0, 0], [0, 0, 16, 0], [32, 0, 0, 64]])  # SURVEY App. A.2 probe
    expect = {
        0: [[2, 4, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0], [32, 64, 0, 0]],
        1: [[2, 8, 16, 4], [32, 0, 0, 64], [0, 0, 0, 0], [0, 0, 0, 0]],
        2: [[0, 0, 2, 4], [0, 0, 0, 8], [0, 0, 0, 16], [0, 0, 32, 64]],
        3: [[0, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 4], [32, 8, 16, 64]],
    }
    for action, want in expect.items():
        got, reward = latent_state(board, action)
        np.testing.assert_array_equal(got, want)
        assert reward == 0
    np.testing.assert_array_equal(board, [[2, 0, 0, 4], [0, 8, 0, 0], [0, 0, 16, 0], [32, 0, 0, 64]])  # input untouched


def test_next_state_valid_and_invalid():
    board = np.array([[2, 2, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])
    new_board, reward = next_state(board, TwentyFortyEight.ACTIONS['left'])  # tests/test_board.py:38-43
    assert new_board[0, 0] == 4 and reward > 0
    new_board, reward = next_state(board, 0, seed=42)  # tests/test_environment.py:138-149
    assert reward == 4 and np.count_nonzero(new_board) == 2
    again, _ = next_state(board, 0, seed=42)
    np.testing.assert_array_equal(new_board, again)  # a seed pins the spawn
    stuck = np.array(LEFT_COLUMN)
    out, reward = next_state(stuck, 0)  # tests/test_environment.py:121-136, tests/test_board.py:45-51
    assert out is stuck and reward == 0 and isinstance(reward, int)  # same object, int 0 (gameboard.py:325)
    for action in range(4):  # tests/test_environment.py:151-164
        result, reward = next_state(board.copy(), action, seed=action)
        assert result.shape == (4, 4)
        if action == 0:
            assert reward > 0


def test_is_done_kats():
    assert is_done(np.array(STAIRCASE))  # tests/test_board.py:53-56
    assert not is_done(np.array([[2, 0, 0, 0], [2, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]))  # :58-61
    one_empty = np.array([[2, 4, 8, 0], [16, 32, 64, 128], [256, 512, 1024, 2048], [4096, 8192, 16384, 32768]])
    assert not is_done(one_empty)  # tests/test_environment.py:231-236
    pair = np.array([[2, 2, 4, 8], [16, 32, 64, 128], [256, 512, 1024, 2048], [4096, 8192, 16384, 32768]])
    assert not is_done(pair)  # :238-243
    assert not is_done(np.zeros((4, 4), dtype=np.int64))  # NumPy semantics for the empty board (App. A.5)


def test_legal_and_illegal_actions():
    board = np.array([[2, 0, 0, 0], [2, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])  # tests/test_move.py:13-23
    assert set(illegal_actions(board)) == {0}
    assert set(legal_actions(board)) == {1, 2, 3}
    spread = np.zeros((4, 4), dtype=int)
    spread[0, 0] = 2
    spread[1, 1] = 2
    assert len(legal_actions(spread)) == 4  # tests/test_environment.py:166-173
    stuck = np.array(LEFT_COLUMN)
    assert 0 in illegal_actions(stuck) and len(legal_actions(stuck)) + len(illegal_actions(stuck)) == 4
    assert legal_actions_mask(stuck) == (False, False, True, False)  # a full column of distinct tiles only moves right
    assert not can_move(stuck) and can_move(np.rot90(stuck, k=2))


def test_after_state_enumeration():
    board = np.array([[2, 4, 0, 0], [0, 2, 0, 0], [0, 0, 0, 0], [0, 0, 0, 2]])  # tests/test_board.py:78-104
    eager = after_state(board)
    base, cells, num_empty = after_state_lazy(board)
    assert base is board and num_empty == 12 and len(eager) == 24
    lazy = [generate_outcome(base, cell, value, num_empty) for cell in cells for value in (2, 4)]
    for (e_state, e_prob), (l_state, l_prob) in zip(eager, lazy):  # same order as the reference enumerates
        np.testing.assert_array_equal(e_state, l_state)
        assert e_prob == pytest.approx(l_prob, abs=1e-12)
    assert sum(p for _, p in eager) == pytest.approx(1.0, abs=1e-10)  # tests/test_environment.py:281-289
    single = np.array([[2, 4, 8, 16], [32, 64, 128, 0], [256, 512, 1024, 2048], [4096, 8192, 16384, 32768]])
    outcomes = after_state(single)  # :291-302
    assert sorted(p for _, p in outcomes) == pytest.approx([0.1, 0.9])
    full = np.array([[2, 4, 8, 16], [32, 64, 128, 256], [512, 1024, 2048, 4096], [2, 4, 8, 16]])
    _, cells, num_empty = after_state_lazy(full)  # tests/test_board.py:106-113
    assert num_empty == 0 and cells == []
    assert after_state(full)[0][1] == 1.0
    with pytest.raises(ValueError):
        generate_outcome(full, (0, 0), 2, 0)


def test_after_state_equals_the_reference_fixture(golden):
    """gameboard.after_state against outcomes the reference's own after_state produced (tests/golden/chance_outcomes.npz):
    same order, same boards, the SAME float64 probabilities (the mirror divides in float64 like the reference)."""
    g = golden['chance_outcomes']
    rng = np.random.default_rng(0)
    picks = np.concatenate([rng.choice(g['boards'].shape[0], size=60, replace=False), np.flatnonzero(g['count'] == 1)[:5],
                            np.flatnonzero(g['count'] == 2)[:5], np.flatnonzero(g['count'] == 32)[:2]])
    for i in picks:
        board = oracle.unpack_values(g['boards'][i]).reshape(4, 4)
        outcomes = after_state(board)
        assert len(outcomes) == int(g['count'][i])
        for s, (state, prob) in enumerate(outcomes):
            np.testing.assert_array_equal(oracle.pack_values(state).reshape(16), g['outcomes'][i, s])
            assert prob == g['prob'][i, s]


def test_fill_cells_mutates_in_place():
    state = np.zeros((4, 4), dtype=np.int64)
    out = fill_cells(state=state, number_tile=2, seed=7)
    assert out is state and np.count_nonzero(state) == 2 and set(state[state != 0]) <= {2, 4}
    full = np.array(STAIRCASE)
    assert fill_cells(full, 1, seed=1) is full and np.array_equal(full, STAIRCASE)
    almost = np.array(STAIRCASE)
    almost[2, 2] = 0
    fill_cells(almost, 5, seed=3)  # clamps to the single empty cell
    assert np.count_nonzero(almost) == 16 and almost[2, 2] in (2, 4)


def test_unsupported_sizes_raise():
    """Sizes 2 ... 8 are served (tests/test_nxn.py); beyond that, and non-square boards, are refused loudly."""
    with pytest.raises(ValueError):
        TwentyFortyEight(size=9)
    with pytest.raises(ValueError):
        slide_and_merge(np.zeros((9, 9), dtype=np.int64))
    with pytest.raises(ValueError):
        slide_and_merge(np.zeros((4, 6), dtype=np.int64))
    with pytest.raises(ValueError):
        merge_column(np.array([2] * 9))


# ------------------------------------------------------------------ the env class
def test_env_reset_invariants():
    env = TwentyFortyEight(size=4)
    obs = env.reset()  # tests/test_environment.py:31-46, tests/test_board.py:32-36
    assert np.count_nonzero(obs) == 2 and np.all(np.isin(obs[obs != 0], [2, 4]))
    assert env.reward == 0 and not env.is_finished
    np.testing.assert_array_equal(env.reset(seed=42), env.reset(seed=42))  # :48-53
    assert env._current_state.dtype == np.int64 and env.observation.shape == (4, 4)


def test_env_step_signature_and_modes():
    env = TwentyFortyEight(size=4)
    env.reset()
    result = env.step(0)  # tests/test_environment.py:55-73
    assert isinstance(result, tuple) and len(result) == 3
    obs, reward, done = result
    assert isinstance(obs, np.ndarray) and isinstance(reward, (int, float, np.integer, np.floating))
    assert isinstance(done, (bool, np.bool_))
    encoded = TwentyFortyEight(size=4, encoded=True)  # :87-97
    obs = encoded.reset()
    assert obs.ndim == 1 and len(obs) == 4 * 4 * 31 and obs.sum() == 16
    for action in (0, 1, 2, 3):  # (step + one-hot observation as a launch pair: same as encoding the new board afterwards)
        obs, _, _ = encoded.step(action)
        np.testing.assert_array_equal(obs, encoded.observation)
        assert obs.dtype == np.int64 and obs.sum() == 16
    board = np.array([[2, 2, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])
    env_norm, env_raw = TwentyFortyEight(normalize=True), TwentyFortyEight(normalize=False)
    env_norm._current_state, env_raw._current_state = board.copy(), board.copy()  # :99-115
    _, reward_norm, _ = env_norm.step(0)
    _, reward_raw, _ = env_raw.step(0)
    assert 0 < reward_norm < reward_raw and reward_raw == 4
    assert reward_norm == pytest.approx(np.log2(4) / 16)


def test_env_spawn_distribution():
    counts = {2: 0, 4: 0}
    env = TwentyFortyEight(size=4)
    for i in range(400):  # tests/test_environment.py:249-279 (90/10 within +-5 %)
        env.reset(seed=i)
        tiles = env._current_state[env._current_state != 0]
        assert np.all((tiles == 2) | (tiles == 4))
        for tile in tiles:
            counts[int(tile)] += 1
    total = sum(counts.values())
    assert counts[2] / total == pytest.approx(TILE_SPAWN_PROBS[2], abs=0.05)
    assert counts[4] / total == pytest.approx(TILE_SPAWN_PROBS[4], abs=0.05)


def test_env_plays_to_termination_and_resets():
    env = TwentyFortyEight(size=4)
    for game in range(2):  # tests/test_environment.py:308-365
        env.reset(seed=42 + game)
        for step in range(1000):
            legal = legal_actions(env._current_state)
            if not legal or env.is_finished:
                break
            _, _, done = env.step(legal[0])
            if done:
                break
        assert env.is_finished or step < 999
        assert env.observation is not None
    env._current_state = np.array(STAIRCASE)
    assert env.is_finished
    obs = env.reset()
    assert np.count_nonzero(obs) == 2 and not env.is_finished and env.reward == 0


def test_render_prints_tab_separated_rows(capsys):
    env = TwentyFortyEight()
    env._current_state = np.array(LEFT_COLUMN)
    env.render()
    assert capsys.readouterr().out.splitlines()[0] == '2 \t0 \t0 \t0'


# ------------------------------------------------------------------ post-transforms vs reference vectors
def test_transforms_match_reference_vectors(golden):
    g = golden['transforms']
    import oracle

    for i in range(0, g['boards'].shape[0], 9):
        values = oracle.unpack_values(g['boards'][i]).reshape(4, 4)
        np.testing.assert_array_equal(encode_flatten(values, encodage_size=31), g['onehot31'][i])
    for r, want in zip(g['rewards'], g['normalized']):
        assert normalize_reward(float(r)) == pytest.approx(float(want), abs=1e-12)


def test_single_board_calls_from_several_threads(api_runtime):
    """Every host thread has its own scratch operands (simulate_2048_amd/scalar.py): four threads stepping their own
    environments concurrently get exactly the games they get one after the other."""
    import threading

    def play(seed, out):  # (the runtime selected by the fixture is process-wide: worker threads see it too)
        state = np.zeros((4, 4), dtype=np.int64)
        fill_cells(state, 2, seed=seed)
        rng = np.random.default_rng(seed)
        total = 0.0
        for t in range(60):
            state, r = next_state(state, int(rng.integers(0, 4)), seed=seed * 1000 + t)
            total += float(r)
        out[seed] = (state.copy(), total)

    serial, threaded = {}, {}
    for seed in (1, 2, 3, 4):
        play(seed, serial)
    threads = [threading.Thread(target=play, args=(seed, threaded)) for seed in (1, 2, 3, 4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for seed in (1, 2, 3, 4):
        np.testing.assert_array_equal(threaded[seed][0], serial[seed][0])
        assert threaded[seed][1] == serial[seed][1]
