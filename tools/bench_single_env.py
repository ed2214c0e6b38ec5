#!/usr/bin/env python3
"""Latency of the single-board API mirrors on the GPU box: BASELINE config 0's workload through `TwentyFortyEight.step`, the
core functions, a 6 x 6 env (general kernels), and one move of the reference's SINGLE-game loop on the `reinforce.game`
mirrors (get_observation + get_legal_actions + step with a fresh key, reinforce/training/self_play.py:84-135, evaluate.py:89-128)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd.reinforce.game import env as genv  # noqa: E402
from simulate_2048_amd.twentyfortyeight.core import gameboard  # noqa: E402
from simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight import TwentyFortyEight  # noqa: E402


def play(env, acts):
    for a in acts[:50]:
        env.step(int(a))
    t, n = time.perf_counter(), 0
    for a in acts:
        _, _, d = env.step(int(a))
        n += 1
        if d:
            env.reset()
    dt = time.perf_counter() - t
    return n / dt, dt / n * 1e6


acts = np.random.default_rng(0).integers(0, 4, 1000)
env = TwentyFortyEight()
env.reset(seed=0)
print('TwentyFortyEight.step: %.0f steps/s (%.1f us/step)' % play(env, acts))
env6 = TwentyFortyEight(size=6)
env6.reset(seed=0)
print('TwentyFortyEight(size=6).step: %.0f steps/s (%.1f us/step)' % play(env6, acts))
b = np.array([[2, 2, 0, 0], [0, 4, 4, 0], [0, 0, 8, 8], [2, 0, 0, 2]])
t = time.perf_counter()
for i in range(300):
    gameboard.latent_state(b, i & 3)
print('latent_state: %.1f us/call' % ((time.perf_counter() - t) / 300 * 1e6))
t = time.perf_counter()
for i in range(300):
    gameboard.next_state(b.copy(), i & 3, seed=i)
print('next_state: %.1f us/call' % ((time.perf_counter() - t) / 300 * 1e6))
# one move of the single-game loop on the reinforce.game mirrors: observation, legal mask, step, scalar read-backs
state = genv.reset(7)
moves = 600
for warm in range(2):
    t = time.perf_counter()
    for m in range(moves):
        obs = np.array(genv.get_observation(state))
        legal = np.array(genv.get_legal_actions(state))
        action = int(np.flatnonzero(legal)[0]) if legal.any() else 0
        state, reward, done, info = genv.step(state, action, [m, 123])
        if bool(done):
            state = genv.reset(1000 + m)
        float(reward)
    dt = (time.perf_counter() - t) / moves * 1e6
print('reinforce.game single-game move (get_observation + get_legal_actions + step + read-backs): %.1f us' % dt)
# one move of evaluate_games / play_game on the CORE functions (reinforce/evaluate.py:96-128): encode_observation, legal_actions_mask,
# next_state, is_done, float(reward), _replace
from simulate_2048_amd.reinforce.game.core import encode_observation, is_done, legal_actions_mask, max_tile, next_state  # noqa: E402

state = genv.reset(11)
for warm in range(2):
    t = time.perf_counter()
    for m in range(moves):
        obs = np.array(encode_observation(state.board))
        legal = np.array(legal_actions_mask(state.board))
        action = int(np.flatnonzero(legal)[0]) if legal.any() else 0
        new_board, reward = next_state(state.board, action, [m, 321])
        done = is_done(new_board)
        float(reward)
        state = state._replace(board=new_board, done=done)
        if bool(state.done):
            int(max_tile(state.board))
            state = genv.reset(2000 + m)
    dt = (time.perf_counter() - t) / moves * 1e6
print('reinforce.game evaluate_games move on the core functions (encode_observation + legal_actions_mask + next_state + is_done + read-backs): %.1f us' % dt)
