"""
The ``reinforce.game`` drop-in surface (simulate_2048_amd.reinforce.game.{core,env}), exercised with
the known-answer vectors of the reference's tests/test_jax_game.py (cited per test) plus the
batched / done-guard behaviour that file never pins.  JAX is not installable here, so the reference
side of these vectors is the plain data its tests hold; torch tensors stand in for JAX arrays.
"""

import numpy as np
import pytest
import torch

from simulate_2048_amd.reinforce.game.core import (
    count_empty,
    create_initial_board,
    encode_observation,
    fill_cells,
    is_done,
    latent_state,
    legal_actions_mask,
    max_tile,
    merge_row,
    next_state,
    slide_and_merge,
)
from simulate_2048_amd.reinforce.game.env import (
    Environment,
    GameState,
    batched_get_legal_actions,
    batched_get_observation,
    batched_reset,
    batched_step,
    get_legal_actions,
    get_observation,
    reset,
    step,
)

pytestmark = pytest.mark.usefixtures('api_runtime')

KEY = np.array([0, 42], dtype=np.uint32)  # what jax.random.PRNGKey(42) looks like


def T(x):
    return torch.tensor(x, dtype=torch.int32)


def L(t):
    return t.cpu().tolist()


@pytest.mark.parametrize(
    'row,expect,score',
    [  # tests/test_jax_game.py:35-82
        ([0, 0, 0, 0], [0, 0, 0, 0], 0),
        ([0, 0, 2, 0], [2, 0, 0, 0], 0),
        ([2, 2, 0, 0], [4, 0, 0, 0], 4),
        ([2, 2, 2, 2], [4, 4, 0, 0], 8),
        ([2, 4, 8, 16], [2, 4, 8, 16], 0),
        ([2, 0, 2, 4], [4, 4, 0, 0], 4),
        ([4, 2, 2, 0], [4, 4, 0, 0], 4),
    ],
)
def test_merge_row(row, expect, score):
    result, s = merge_row(T(row))
    assert L(result) == expect and int(s) == score and result.dtype == torch.int32


def test_slide_and_merge_full_board():
    board = T([[2, 2, 4, 4], [4, 4, 2, 2], [2, 4, 2, 4], [4, 2, 4, 2]])  # tests/test_jax_game.py:88-95
    result, score = slide_and_merge(board)
    assert L(result) == [[4, 8, 0, 0], [8, 4, 0, 0], [2, 4, 2, 4], [4, 2, 4, 2]] and int(score) == 24


@pytest.mark.parametrize(
    'action,board,expect',
    [  # tests/test_jax_game.py:101-135
        (0, [[0, 2, 0, 2], [0] * 4, [0] * 4, [0] * 4], [[4, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4]),
        (1, [[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4], [[4, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4]),
        (2, [[2, 0, 0, 2], [0] * 4, [0] * 4, [0] * 4], [[0, 0, 0, 4], [0] * 4, [0] * 4, [0] * 4]),
        (3, [[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4], [[0] * 4, [0] * 4, [0] * 4, [4, 0, 0, 0]]),
    ],
)
def test_latent_state_directions(action, board, expect):
    result, reward = latent_state(T(board), action)
    assert L(result) == expect and int(reward) == 4


def test_legal_actions_mask_and_is_done():
    empty = torch.zeros((4, 4), dtype=torch.int32)  # tests/test_jax_game.py:141-145
    assert L(legal_actions_mask(empty)) == [False] * 4
    assert bool(is_done(empty))  # JAX semantics: no legal action == done (core.py:320-321)
    checker = T([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 2]])  # :147-151, :179-182
    assert L(legal_actions_mask(checker)) == [False] * 4 and bool(is_done(checker))
    almost = T([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 2, 2]])  # :174-177
    assert not bool(is_done(almost))
    single = T([[2, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4])
    mask = legal_actions_mask(single)
    assert mask.dtype == torch.bool and L(mask) == [False, False, True, True]
    assert not bool(is_done(T([[2, 0, 0, 0], [2, 0, 0, 0], [0] * 4, [0] * 4])))  # :169-172


def test_fill_cells():
    board = T([[2, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4])  # tests/test_jax_game.py:188-199
    out = fill_cells(board, KEY)
    assert int((out != 0).sum()) == 2 and int(out[0, 0]) == 2
    new = out.clone()
    new[0, 0] = 0
    assert int(new.max()) in (2, 4)
    full = T([[2, 4, 8, 16], [32, 64, 128, 256], [512, 1024, 2048, 4096], [2, 4, 8, 16]])  # :201-206
    assert torch.equal(fill_cells(full, KEY).cpu(), full)


def test_next_state_up():
    board = T([[2, 0, 0, 0], [0] * 4, [2, 0, 0, 0], [0] * 4])  # tests/test_jax_game.py:212-225
    new_board, reward = next_state(board, 1, KEY)
    assert int(new_board[0, 0]) == 4 and int((new_board != 0).sum()) == 2
    assert float(reward) == 4.0 and reward.dtype == torch.float32
    same, reward = next_state(T([[2, 0, 0, 0], [4, 0, 0, 0], [8, 0, 0, 0], [16, 0, 0, 0]]), 0, KEY)
    assert float(reward) == 0.0 and int((same != 0).sum()) == 4  # invalid move: no spawn (core.py:258-270)


def test_reset_and_step():
    state = reset(KEY)  # tests/test_jax_game.py:231-241
    assert isinstance(state, GameState) and tuple(state.board.shape) == (4, 4) and state.board.dtype == torch.int32
    assert int((state.board != 0).sum()) == 2 and int(state.step_count) == 0
    assert not bool(state.done) and float(state.total_reward) == 0.0
    assert torch.equal(reset(KEY).board, state.board) and torch.equal(create_initial_board(KEY), state.board)
    new_state, reward, done, info = step(state, 0, np.array([0, 43], dtype=np.uint32))  # :243-252
    assert int(new_state.step_count) == 1 and 'max_tile' in info and int(info['step_count']) == 1
    assert float(info['total_reward']) == float(reward) and done.dtype == torch.bool
    obs = get_observation(state)  # :254-261
    assert tuple(obs.shape) == (16,) and obs.dtype == torch.float32
    mask = get_legal_actions(state)  # :263-270
    assert tuple(mask.shape) == (4,) and mask.dtype == torch.bool


def test_encode_max_count():
    board = T([[2, 4, 8, 16], [0] * 4, [0] * 4, [0] * 4])  # tests/test_jax_game.py:282-291
    obs = encode_observation(board)
    assert L(obs[:4]) == [1 / 16, 2 / 16, 3 / 16, 4 / 16] and float(obs[4:].abs().sum()) == 0.0
    assert float(encode_observation(torch.zeros((4, 4), dtype=torch.int32)).abs().sum()) == 0.0  # :276-280
    assert int(max_tile(T([[2, 4, 8, 16], [32, 64, 128, 256], [0] * 4, [0] * 4]))) == 256  # :297-301
    assert int(max_tile(torch.zeros((4, 4), dtype=torch.int32))) == 0
    assert int(count_empty(torch.zeros((4, 4), dtype=torch.int32))) == 16  # :307-317
    assert int(count_empty(T([[2, 4, 8, 16], [32, 64, 128, 256], [0] * 4, [0] * 4]))) == 8


def test_batched_api_shapes_and_values():
    n = 37
    keys = np.stack([np.zeros(n, dtype=np.uint32), np.arange(n, dtype=np.uint32) + 5], axis=1)
    states = batched_reset(keys)  # tests/test_jax_game.py:356-374
    assert tuple(states.board.shape) == (n, 4, 4) and tuple(states.done.shape) == (n,)
    assert torch.all((states.board != 0).sum(dim=(1, 2)) == 2)
    obs, legal = batched_get_observation(states), batched_get_legal_actions(states)
    assert tuple(obs.shape) == (n, 16) and tuple(legal.shape) == (n, 4) and legal.dtype == torch.bool
    np.testing.assert_array_equal(obs.cpu().numpy() * 16, np.log2(np.maximum(states.board.cpu().numpy().reshape(n, 16), 1)))
    actions = np.arange(n) % 4
    new_states, rewards, dones, info = batched_step(states, actions, keys + 1)
    assert rewards.dtype == torch.float32 and dones.dtype == torch.bool and tuple(rewards.shape) == (n,)
    assert torch.equal(new_states.step_count.cpu(), torch.ones(n, dtype=torch.int32))  # +1 also on invalid moves (env.py:116)
    assert torch.equal(info['max_tile'].cpu(), new_states.board.reshape(n, 16).max(dim=1).values.cpu())
    # every env: either unchanged (invalid move, reward 0) or exactly one more/equal tile count with merges
    moved = (new_states.board != states.board).reshape(n, 16).any(dim=1)
    was_legal = legal.cpu()[torch.arange(n), torch.as_tensor(actions)]
    assert torch.equal(moved.cpu(), was_legal)
    assert torch.all(rewards[~moved] == 0)


def test_done_guard_freezes_finished_games():
    """reinforce/game/env.py:103-104: a done state is returned unchanged with reward 0.0, done True."""
    checker = T([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 2]])
    live = T([[2, 2, 0, 0], [0] * 4, [0] * 4, [0] * 4])
    states = GameState(
        board=torch.stack([checker, live, live]),
        step_count=torch.tensor([7, 3, 9], dtype=torch.int32),
        done=torch.tensor([True, False, True]),  # third: flagged done although its board could move
        total_reward=torch.tensor([100.0, 8.0, 5.0]),
    )
    new_states, rewards, dones, _ = batched_step(states, [0, 0, 0], np.array([[0, 1], [0, 2], [0, 3]], dtype=np.uint32))
    assert L(new_states.step_count) == [7, 4, 9] and L(dones) == [True, False, True]
    assert L(rewards) == [0.0, 4.0, 0.0] and L(new_states.total_reward) == [100.0, 12.0, 5.0]
    assert torch.equal(new_states.board[0].cpu(), checker) and torch.equal(new_states.board[2].cpu(), live)
    assert int(new_states.board[1, 0, 0]) == 4


def test_environment_wrapper():
    env = Environment(seed=3)  # reinforce/game/env.py:248-304
    assert int((env.board != 0).sum()) == 2 and not env.done
    state, reward, done, info = env.step_env(int(torch.nonzero(env.legal_actions())[0]))
    assert isinstance(reward, float) and isinstance(done, bool) and int(state.step_count) == 1
    assert tuple(env.observation().shape) == (16,)
    other = Environment(seed=3)
    assert torch.equal(other.reset_env().board.cpu(), Environment(seed=3).reset_env().board.cpu())
