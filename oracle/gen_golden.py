#!/usr/bin/env python3
"""
Generate tests/golden/*.npz from the REAL reference NumPy engine.

Run in the build container only (the reference lives at /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py

Every expected output stored here is produced by calling the reference's own functions
(``twentyfortyeight.core.gameboard`` / ``gamemove`` / ``utils``); this script only chooses the
inputs and, for the stochastic part, injects the spawn choice (k-th empty cell, value) through
the reference's ``after_state_lazy`` + ``generate_outcome`` so that the build's Philox contract
can be replayed through the reference engine.  Fixtures are data (inputs + expected outputs),
stored as packed log2 exponents (uint8) unless noted.

Fixture sets:
  rows.npz          G1  all 18^4 rows e in [0,17]: merge_column result (padded), score, can_move
  afterstates.npz   G2  boards x 4 actions: latent_state, reward, validity; legal mask, is_done
  next_state.npz    G3  real next_state(board, action, seed=numpy seed): board', reward, inferred (k, e), done
  trajectories.npz  G4  full games under the build's Philox contract replayed by the reference engine
  resets.npz        G5  reset boards under the Philox contract (two sequential injected spawns)
  transforms.npz        encode_flatten(...,31) one-hot observations and normalize_reward values
  chance_outcomes.npz   after_state(board): every (empty cell, value) outcome in the reference's order + float64 probabilities
  expectimax_d2.npz     depth-2 expectimax (Q2, exact integer terms, best action) composed from latent_state / after_state / can_move
  nxn.npz               boards of sizes 2 ... 8: latent_state / can_move / legal mask / is_done / merge_column / after_state and
                        Philox-driven whole games replayed by the reference's functions (the `size` argument of the env)
  random_play_stats.npz per-game length / total reward / max tile of 3000 uniformly-random games played by the REFERENCE engine
                        with its own NumPy RNG: the distribution the build's Philox-driven play must reproduce
"""

from __future__ import annotations

import hashlib
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from twentyfortyeight.core import gameboard as ref_board  # noqa: E402  (reference, read-only)
from twentyfortyeight.core import gamemove as ref_move  # noqa: E402
from twentyfortyeight.utils.binary import encode_flatten as ref_encode_flatten  # noqa: E402
from twentyfortyeight.utils.normalize import normalize_reward as ref_normalize_reward  # noqa: E402

import oracle  # noqa: E402  (only its Philox contract is used here, to choose injected spawns)

OUT = REPO / 'tests' / 'golden'
MAX_E = 17  # largest reachable tile on 4x4 is 2^17


def to_values(exps: np.ndarray) -> np.ndarray:
    e = np.asarray(exps, dtype=np.int64)
    return np.where(e > 0, np.int64(1) << e, 0).astype(np.int64)


def to_exps(values: np.ndarray) -> np.ndarray:
    v = np.asarray(values, dtype=np.int64)
    out = np.zeros(v.shape, dtype=np.uint8)
    nz = v > 0
    out[nz] = np.round(np.log2(v[nz])).astype(np.uint8)
    assert np.all(to_values(out) == v)
    return out


# ------------------------------------------------------------------------------------------------
# G1: exhaustive rows
# ------------------------------------------------------------------------------------------------
def gen_rows():
    base = MAX_E + 1
    n = base**4
    out_rows = np.zeros((n, 4), dtype=np.uint8)
    scores = np.zeros(n, dtype=np.int32)
    can_move = np.zeros(n, dtype=np.uint8)
    for idx in range(n):
        digits = [(idx // base**3) % base, (idx // base**2) % base, (idx // base) % base, idx % base]
        row = to_values(np.array(digits))
        score, merged = ref_board.merge_column(row)
        padded = np.zeros(4, dtype=np.int64)
        padded[: len(merged)] = merged
        out_rows[idx] = to_exps(padded)
        scores[idx] = int(score)
        can_move[idx] = int(ref_move.can_move(row.reshape(1, 4)))
    np.savez_compressed(OUT / 'rows.npz', max_e=np.int32(MAX_E), merged=out_rows, score=scores, can_move=can_move)
    return n


# ------------------------------------------------------------------------------------------------
# Board mixture
# ------------------------------------------------------------------------------------------------
def random_play_boards(rng: np.random.Generator, count: int) -> list[np.ndarray]:
    """States visited by the reference engine itself under uniform random actions."""
    boards = []
    seed = 10_000
    while len(boards) < count:
        state = np.zeros((4, 4), dtype=np.int64)
        state = ref_board.fill_cells(state=state, number_tile=2, seed=seed)
        seed += 1
        for _ in range(600):
            if rng.random() < 0.12:
                boards.append(state.copy())
            if ref_board.is_done(state):
                boards.append(state.copy())
                break
            state, _ = ref_board.next_state(state.copy(), int(rng.integers(0, 4)), seed=seed)
            seed += 1
    return boards[:count]


def mixture(rng: np.random.Generator, n_play: int, n_uniform: int, n_dense: int, n_big: int) -> np.ndarray:
    boards = [to_exps(b).reshape(16) for b in random_play_boards(rng, n_play)]
    for _ in range(n_uniform):  # uniform small exponents, many empties
        boards.append(rng.integers(0, 7, size=16).astype(np.uint8))
    for _ in range(n_dense):  # dense / near-terminal: few distinct exponents, 0-2 empties
        b = rng.integers(1, 5, size=16).astype(np.uint8)
        for _ in range(int(rng.integers(0, 3))):
            b[int(rng.integers(0, 16))] = 0
        boards.append(b)
    for _ in range(n_big):  # big tiles up to 2^17
        b = rng.integers(0, MAX_E + 1, size=16).astype(np.uint8)
        b[rng.random(16) < 0.3] = 0
        boards.append(b)
    hand = [
        np.zeros(16, dtype=np.uint8),  # all-zero: NOT done in the NumPy engine
        np.arange(1, 17, dtype=np.uint8),  # staircase 2..65536: done (tests/test_board.py:53-56)
        np.array([1, 2] * 8, dtype=np.uint8).reshape(4, 4).reshape(16),
        to_exps(np.array([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 2]])).reshape(16),  # checkerboard
        to_exps(np.array([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 2, 2]])).reshape(16),
        np.full(16, 16, dtype=np.uint8),  # all 65536: every move merges to 2^17
        np.full(16, 17, dtype=np.uint8),
        to_exps(np.array([[2, 0, 0, 0], [2, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])).reshape(16),
    ]
    boards.extend(hand)
    return np.stack(boards).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# G2: afterstates + legality + terminal
# ------------------------------------------------------------------------------------------------
def gen_afterstates(rng):
    boards = mixture(rng, n_play=3000, n_uniform=1200, n_dense=1200, n_big=600)
    n = boards.shape[0]
    after = np.zeros((n, 4, 16), dtype=np.uint8)
    reward = np.zeros((n, 4), dtype=np.int32)
    valid = np.zeros((n, 4), dtype=np.uint8)
    legal = np.zeros(n, dtype=np.uint8)
    done = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        state = to_values(boards[i]).reshape(4, 4)
        for a in range(4):
            new_state, r = ref_board.latent_state(state, a)
            after[i, a] = to_exps(new_state).reshape(16)
            reward[i, a] = int(r)
            valid[i, a] = int(ref_move.can_move(np.rot90(state, k=a)))  # the test next_state applies (:316-317)
        mask = ref_move.legal_actions_mask(state)
        legal[i] = sum(int(bool(m)) << a for a, m in enumerate(mask))
        assert set(ref_move.legal_actions(state)) == {a for a in range(4) if mask[a]}
        assert set(ref_move.illegal_actions(state)) == {a for a in range(4) if not mask[a]}
        done[i] = int(ref_board.is_done(state))
    np.savez_compressed(OUT / 'afterstates.npz', boards=boards, after=after, reward=reward, valid=valid,
                        legal=legal, done=done)
    return n


# ------------------------------------------------------------------------------------------------
# G3: the reference's real next_state with its own (NumPy PCG64) seeded spawn
# ------------------------------------------------------------------------------------------------
def gen_next_state(rng):
    boards = mixture(rng, n_play=2400, n_uniform=600, n_dense=800, n_big=200)
    n = boards.shape[0]
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    seeds = rng.integers(1, 2**31 - 1, size=n).astype(np.int64)
    out = np.zeros((n, 16), dtype=np.uint8)
    reward = np.zeros(n, dtype=np.int32)
    spawn_k = np.full(n, 255, dtype=np.uint8)  # 255 = no spawn (invalid move)
    spawn_e = np.zeros(n, dtype=np.uint8)
    done = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        state = to_values(boards[i]).reshape(4, 4)
        new_state, r = ref_board.next_state(state.copy(), int(actions[i]), seed=int(seeds[i]))
        out[i] = to_exps(new_state).reshape(16)
        reward[i] = int(r)
        done[i] = int(ref_board.is_done(new_state))
        after, _ = ref_board.latent_state(state, int(actions[i]))
        diff = np.argwhere(np.asarray(new_state) != np.asarray(after))
        if ref_move.can_move(np.rot90(state, k=int(actions[i]))):
            assert len(diff) == 1, (state, actions[i], new_state)
            cell = (int(diff[0][0]), int(diff[0][1]))
            _, cells, _ = ref_board.after_state_lazy(after)
            spawn_k[i] = cells.index(cell)
            spawn_e[i] = int(np.log2(new_state[cell]))
        else:
            assert len(diff) == 0 and int(r) == 0 and np.array_equal(new_state, state)
    np.savez_compressed(OUT / 'next_state.npz', boards=boards, actions=actions, numpy_seeds=seeds, out=out,
                        reward=reward, spawn_k=spawn_k, spawn_e=spawn_e, done=done)
    return n


# ------------------------------------------------------------------------------------------------
# Reference engine driven by the build's Philox contract (injected spawns)
# ------------------------------------------------------------------------------------------------
def ref_spawn(state: np.ndarray, r_cell: int, r_val: int | None = None) -> np.ndarray:
    """One spawn through after_state_lazy/generate_outcome with (k, e) from Philox words:
    two words (reset block) or, with r_val None, the one-word step form."""
    base, cells, n_empty = ref_board.after_state_lazy(state)
    if n_empty == 0:
        return state
    if r_val is None:
        k, e, _ = oracle.spawn_from_word(r_cell, n_empty)
    else:
        k, e = oracle.spawn_from_words(r_cell, r_val, n_empty)
    new_state, _ = ref_board.generate_outcome(base, cells[k], 1 << e, n_empty)
    return new_state


def ref_reset(seed: int, env: int, step: int, size: int = 4) -> np.ndarray:
    r = oracle.env_block(seed, env, step, 1)
    state = np.zeros((size, size), dtype=np.int64)
    state = ref_spawn(state, int(r[0]), int(r[1]))
    state = ref_spawn(state, int(r[2]), int(r[3]))
    return state


def ref_step(state: np.ndarray, action: int, word: int) -> tuple[np.ndarray, int]:
    """next_state's control flow (gameboard.py:316-325) with the spawn injected."""
    rotated = np.rot90(state, k=action)
    if ref_move.can_move(rotated):
        reward, updated = ref_board.slide_and_merge(rotated)
        new_state = np.rot90(updated, k=-action)
        new_state = ref_spawn(new_state, word)
        return new_state, int(reward)
    return state, 0


def gen_trajectories():
    sets = {}
    for name, seed, env_base, n_env, T, auto_reset, step0 in [
        ('auto', 0x2048, 1001, 40, 420, 1, 7),  # odd env_base: Philox groups straddle the batch edge
        ('plain', 0xC0FFEE, (1 << 32) + 5, 16, 420, 0, 0),  # env index above 2^32 exercises env_hi
    ]:
        boards0 = np.zeros((n_env, 16), dtype=np.uint8)
        boards = np.zeros((T, n_env, 16), dtype=np.uint8)
        actions = np.zeros((T, n_env), dtype=np.uint8)
        reward = np.zeros((T, n_env), dtype=np.int32)
        done = np.zeros((T, n_env), dtype=np.uint8)
        legal = np.zeros((T, n_env), dtype=np.uint8)
        for j in range(n_env):
            env = env_base + j
            state = ref_reset(seed, env, step0)
            boards0[j] = to_exps(state).reshape(16)
            for t in range(T):
                step = step0 + t
                word = oracle.step_word(seed, env, step)
                a = word & 3
                state, r = ref_step(state, a, word)
                d = bool(ref_board.is_done(state))
                if d and auto_reset:
                    state = ref_reset(seed, env, step)
                actions[t, j] = a
                reward[t, j] = r
                done[t, j] = int(d)
                boards[t, j] = to_exps(state).reshape(16)
                mask = ref_move.legal_actions_mask(state)
                legal[t, j] = sum(int(bool(m)) << b for b, m in enumerate(mask))
        sets[name] = dict(seed=np.uint64(seed), env_base=np.uint64(env_base), step0=np.uint64(step0),
                          auto_reset=np.uint8(auto_reset), boards0=boards0, boards=boards, actions=actions,
                          reward=reward, done=done, legal=legal)
    flat = {f'{k}_{f}': v for k, d in sets.items() for f, v in d.items()}
    np.savez_compressed(OUT / 'trajectories.npz', **flat)
    return {k: (int(d['done'].sum()), d['boards'].shape) for k, d in sets.items()}


def gen_resets():
    seed, step, env_base, n = 0xABCDEF0123, 3, 77, 4096
    boards = np.zeros((n, 16), dtype=np.uint8)
    for j in range(n):
        boards[j] = to_exps(ref_reset(seed, env_base + j, step)).reshape(16)
    np.savez_compressed(OUT / 'resets.npz', seed=np.uint64(seed), step=np.uint64(step),
                        env_base=np.uint64(env_base), boards=boards)
    return n


def gen_transforms(rng):
    boards = mixture(rng, n_play=200, n_uniform=20, n_dense=20, n_big=16)
    n = boards.shape[0]
    onehot = np.zeros((n, 16 * 31), dtype=np.uint8)
    for i in range(n):
        onehot[i] = ref_encode_flatten(to_values(boards[i]).reshape(4, 4), encodage_size=31).astype(np.uint8)
    rewards = np.array([0, 4, 8, 12, 16, 20, 28, 60, 64, 128, 2048, 4100, 65536, 131072, 2**20], dtype=np.int64)
    norm = np.array([float(ref_normalize_reward(float(r))) for r in rewards], dtype=np.float64)
    np.savez_compressed(OUT / 'transforms.npz', boards=boards, onehot31=onehot, rewards=rewards,
                        normalized=norm)
    return n


def gen_chance_outcomes():
    """The reference's after_state (gameboard.py:132-171) itself: enumeration order and float64 probabilities.
    Own generator, so the older fixture sets keep their bytes."""
    rng = np.random.default_rng(20481)
    boards = mixture(rng, n_play=420, n_uniform=60, n_dense=80, n_big=40)
    extra = []
    for _ in range(60):  # exactly one empty cell / full boards / one tile only
        b = rng.integers(1, 12, size=16).astype(np.uint8)
        if rng.random() < 0.6:
            b[int(rng.integers(0, 16))] = 0
        extra.append(b)
    for c in range(16):
        b = np.zeros(16, dtype=np.uint8)
        b[c] = int(rng.integers(1, MAX_E + 1))
        extra.append(b)
    boards = np.concatenate([boards, np.stack(extra)]).astype(np.uint8)
    n = boards.shape[0]
    out = np.zeros((n, 32, 16), dtype=np.uint8)
    prob = np.zeros((n, 32), dtype=np.float64)
    count = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        outcomes = ref_board.after_state(to_values(boards[i]).reshape(4, 4))
        count[i] = len(outcomes)
        for s, (state, pr) in enumerate(outcomes):
            out[i, s] = to_exps(state).reshape(16)
            prob[i, s] = pr
    np.savez_compressed(OUT / 'chance_outcomes.npz', boards=boards, outcomes=out, prob=prob, count=count)
    return n, int(count.max()), int((count == 1).sum())


def gen_expectimax_d2():
    """Depth-2 expectimax composed LITERALLY from the reference's functions on the chance_outcomes boards:
    Q2(s, a) = latent_state reward + sum over after_state(afterstate) outcomes of prob * max over legal a' of the latent_state
    reward of the outcome (legal = can_move of the rotated board, as next_state decides it, gameboard.py:316-317); float64,
    summed in after_state's order.  Pins oracle_expectimax_depth2 / env2048_expectimax_depth2 (exact integer terms; q to 1e-12)."""
    boards = np.load(OUT / 'chance_outcomes.npz')['boards']
    n = boards.shape[0]
    q = np.full((n, 4), -np.inf, dtype=np.float64)
    terms = np.zeros((n, 4, 4), dtype=np.int32)
    best = np.full(n, 255, dtype=np.uint8)
    value = np.zeros(n, dtype=np.float64)

    def legal(state, a):
        return bool(ref_move.can_move(np.rot90(state, k=a)))

    def best_reward(state):
        rewards = [float(ref_board.latent_state(state, a2)[1]) for a2 in range(4) if legal(state, a2)]
        return max(rewards) if rewards else 0.0

    for i in range(n):
        state = to_values(boards[i]).reshape(4, 4)
        for a in range(4):
            if not legal(state, a):
                continue
            after, r = ref_board.latent_state(state, a)
            outcomes = ref_board.after_state(after)
            expect, s2, s4 = 0.0, 0, 0
            for o, pr in outcomes:
                v = best_reward(o)
                expect += pr * v
                spawned = int((o != after).sum()) and int(o[o != after][0])
                if spawned == 2:
                    s2 += int(v)
                else:
                    s4 += int(v)
            q[i, a] = float(r) + expect
            terms[i, a] = (int(r), s2, s4, int((after == 0).sum()))
        if np.isfinite(q[i]).any():
            best[i] = int(np.argmax(q[i]))  # first maximum
            value[i] = q[i, best[i]]
    np.savez_compressed(OUT / 'expectimax_d2.npz', boards=boards, q=q, terms=terms, best=best, value=value)
    return n, int((best == 255).sum()), float(np.nanmax(np.where(np.isfinite(q), q, np.nan)))


def _ref_legal(state, a):
    return bool(ref_move.can_move(np.rot90(state, k=a)))


def _ref_value(state, depth):
    """V_depth(state) by plain recursion over the reference's own functions (see gen_expectimax_d3)."""
    best = None
    for a in range(4):
        if not _ref_legal(state, a):
            continue
        after, r = ref_board.latent_state(state, a)
        q = float(r)
        if depth > 1:
            q += sum(pr * _ref_value(o, depth - 1) for o, pr in ref_board.after_state(after))
        best = q if best is None or q > best else best
    return 0.0 if best is None else best


def _ref_q3(exps):
    """(q (4,), best, value) of one board at depth 3."""
    state = to_values(exps).reshape(4, 4)
    q = np.full(4, -np.inf, dtype=np.float64)
    for a in range(4):
        if _ref_legal(state, a):
            after, r = ref_board.latent_state(state, a)
            q[a] = float(r) + sum(pr * _ref_value(o, 2) for o, pr in ref_board.after_state(after))
    if np.isfinite(q).any():
        best = int(np.argmax(q))  # first maximum
        return q, best, q[best]
    return q, 255, 0.0


def _ref_q(exps, depth):
    """(q (4,), best, value) of one board at `depth` by the same recursion."""
    state = to_values(exps).reshape(4, 4)
    q = np.full(4, -np.inf, dtype=np.float64)
    for a in range(4):
        if _ref_legal(state, a):
            after, r = ref_board.latent_state(state, a)
            q[a] = float(r) + sum(pr * _ref_value(o, depth - 1) for o, pr in ref_board.after_state(after))
    if np.isfinite(q).any():
        best = int(np.argmax(q))
        return q, best, q[best]
    return q, 255, 0.0


def _ref_q4(exps):
    return _ref_q(exps, 4)


def gen_expectimax_d4(count: int = 24):
    """Depth FOUR by the same literal recursion over the reference's functions, on the boards of the chance_outcomes set with the
    fewest empty cells (1 ... 4: a depth-4 tree of a mid-game board has 10^6 leaves and would take minutes per board in Python;
    these take seconds) -- pins the oracle's recursion, and through it the device's expand -> depth-2 -> back-up chain, one
    level deeper than expectimax_d3.npz."""
    import multiprocessing as mp
    import os

    boards = np.load(OUT / 'chance_outcomes.npz')['boards']
    empties = (boards == 0).sum(axis=1)
    pick = [int(i) for e in (1, 2, 3, 4) for i in np.flatnonzero(empties == e)[:count // 4]]  # six boards per number of empty cells
    boards = boards[pick]
    with mp.get_context('fork').Pool(min(8, os.cpu_count() or 1)) as pool:
        results = pool.map(_ref_q4, [boards[i] for i in range(boards.shape[0])], chunksize=1)
    q = np.stack([r[0] for r in results]).astype(np.float64)
    best = np.array([r[1] for r in results], dtype=np.uint8)
    val = np.array([r[2] for r in results], dtype=np.float64)
    np.savez_compressed(OUT / 'expectimax_d4.npz', boards=boards, q=q, best=best, value=val)
    return boards.shape[0], int((best == 255).sum()), float(val.max())


def gen_expectimax_d3(count: int = 512):
    """Depth-3 expectimax composed LITERALLY from the reference's functions (latent_state, after_state with ITS probabilities,
    can_move), by plain recursion, float64, outcomes in after_state's order: Q3(s, a) = r + sum P(o) * V2(o),
    V2(o) = max over legal a' of [r' + sum P(o') * best legal latent_state reward of o'].  Pins oracle_expectimax_depth /
    the device back-up chain to ~1e-12 relative (the build sums 0.9 * S2 + 0.1 * S4 and divides once; the reference's
    per-outcome products round differently) and the arg-max wherever it is not a near-tie.  512 boards spread evenly over the
    chance_outcomes set (random-play states of all depths, terminal boards included); about two seconds of reference time per
    board, so the boards are farmed out to a process pool (the result does not depend on the pool: one board per task)."""
    import multiprocessing as mp
    import os

    boards = np.load(OUT / 'chance_outcomes.npz')['boards']
    pick = np.linspace(0, boards.shape[0] - 1, count).astype(int)
    boards = boards[pick]
    n = boards.shape[0]
    with mp.get_context('fork').Pool(min(8, os.cpu_count() or 1)) as pool:
        results = pool.map(_ref_q3, [boards[i] for i in range(n)], chunksize=4)
    q = np.stack([r[0] for r in results]).astype(np.float64)
    best = np.array([r[1] for r in results], dtype=np.uint8)
    val = np.array([r[2] for r in results], dtype=np.float64)
    np.savez_compressed(OUT / 'expectimax_d3.npz', boards=boards, q=q, best=best, value=val)
    return n, int((best == 255).sum()), float(val.max())


NXN_SIZES = (2, 3, 5, 6, 8, 7, 4)  # (7 and 4 appended in round 5: the generator is shared, so the earlier sizes keep their data)


def gen_nxn():
    """Boards of other sizes than 4 (the reference's functions take any shape; its env has a `size` argument,
    twentyfortyeight.py:21, and its timing script runs 4 / 6 / 8): per size S -- states of the reference engine's own random
    play + random / dense boards -> latent_state x 4 (board, reward), can_move(rot90) x 4, legal_actions_mask, is_done;
    merge_column on random lines of length S; after_state outcomes of a few boards; and whole games driven by the build's
    Philox contract through the reference's functions (reset, step with injected spawn, auto-reset), as in trajectories.npz."""
    rng = np.random.default_rng(2048_8)
    out = {'sizes': np.array(NXN_SIZES, dtype=np.int32)}
    total = 0
    for S in NXN_SIZES:
        cells = S * S
        boards = []
        seed = 77_000 + 1000 * S
        while len(boards) < 150:  # the reference engine's own random play at this size
            state = ref_board.fill_cells(state=np.zeros((S, S), dtype=np.int64), number_tile=2, seed=seed)
            seed += 1
            for _ in range(400):
                if rng.random() < 0.1:
                    boards.append(to_exps(state).reshape(cells))
                if ref_board.is_done(state):
                    boards.append(to_exps(state).reshape(cells))
                    break
                state, _ = ref_board.next_state(state.copy(), int(rng.integers(0, 4)), seed=seed)
                seed += 1
        boards = boards[:150]
        for _ in range(80):  # uniform small exponents, many empties
            boards.append(rng.integers(0, 6, size=cells).astype(np.uint8))
        for _ in range(80):  # dense: few distinct exponents, 0-2 empties (merges everywhere, near-terminal)
            b = rng.integers(1, 4, size=cells).astype(np.uint8)
            for _ in range(int(rng.integers(0, 3))):
                b[int(rng.integers(0, cells))] = 0
            boards.append(b)
        checker = (np.indices((S, S)).sum(axis=0) % 2 + 1).astype(np.uint8).reshape(cells)
        boards += [np.zeros(cells, dtype=np.uint8), checker, np.full(cells, 3, dtype=np.uint8)]
        boards = np.stack(boards).astype(np.uint8)
        n = boards.shape[0]
        after = np.zeros((n, 4, cells), dtype=np.uint8)
        reward = np.zeros((n, 4), dtype=np.int32)
        valid = np.zeros((n, 4), dtype=np.uint8)
        legal = np.zeros(n, dtype=np.uint8)
        done = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            state = to_values(boards[i]).reshape(S, S)
            for a in range(4):
                new_state, r = ref_board.latent_state(state, a)
                after[i, a] = to_exps(new_state).reshape(cells)
                reward[i, a] = int(r)
                valid[i, a] = int(ref_move.can_move(np.rot90(state, k=a)))
            mask = ref_move.legal_actions_mask(state)
            legal[i] = sum(int(bool(m)) << a for a, m in enumerate(mask))
            done[i] = int(ref_board.is_done(state))
        lines = (rng.integers(0, 5, size=(400, S)) * (rng.random((400, S)) < 0.75)).astype(np.uint8)
        merged = np.zeros((400, S), dtype=np.uint8)
        merged_len = np.zeros(400, dtype=np.uint8)
        line_score = np.zeros(400, dtype=np.int32)
        for i in range(400):
            sc, m = ref_board.merge_column(to_values(lines[i]))
            merged[i, :len(m)] = to_exps(np.asarray(m))
            merged_len[i] = len(m)
            line_score[i] = int(sc)
        # after_state outcomes of 12 boards (incl. the full checkerboard): the reference's enumeration order
        pick = list(range(0, 140, 14)) + [n - 2, n - 3]
        outcomes = np.zeros((len(pick), 2 * cells, cells), dtype=np.uint8)
        outcome_count = np.zeros(len(pick), dtype=np.uint8)
        for j, i in enumerate(pick):
            res = ref_board.after_state(to_values(boards[i]).reshape(S, S))
            outcome_count[j] = len(res)
            for t, (o, _) in enumerate(res):
                outcomes[j, t] = to_exps(o).reshape(cells)
        # whole games under the build's Philox contract, replayed by the reference's functions
        n_env, T, tseed, env_base, step0 = 6, 160, 0x5000 + S, 101 + S, 3
        t_boards = np.zeros((T + 1, n_env, cells), dtype=np.uint8)
        t_actions = np.zeros((T, n_env), dtype=np.uint8)
        t_reward = np.zeros((T, n_env), dtype=np.int32)
        t_done = np.zeros((T, n_env), dtype=np.uint8)
        for j in range(n_env):
            env = env_base + j
            state = ref_reset(tseed, env, step0, S)
            t_boards[0, j] = to_exps(state).reshape(cells)
            for t in range(T):
                word = oracle.step_word(tseed, env, step0 + 1 + t)
                action = int(rng.integers(0, 4))
                state, r = ref_step(state, action, word)
                finished = bool(ref_board.is_done(state))
                if finished:  # auto-reset: a fresh board from the reset block of the same (env, step)
                    state = ref_reset(tseed, env, step0 + 1 + t, S)
                t_actions[t, j], t_reward[t, j], t_done[t, j] = action, int(r), int(finished)
                t_boards[t + 1, j] = to_exps(state).reshape(cells)
        out.update({f's{S}_boards': boards, f's{S}_after': after, f's{S}_reward': reward, f's{S}_valid': valid, f's{S}_legal': legal,
                    f's{S}_done': done, f's{S}_lines': lines, f's{S}_merged': merged, f's{S}_merged_len': merged_len,
                    f's{S}_line_score': line_score, f's{S}_outcome_pick': np.array(pick, dtype=np.int32), f's{S}_outcomes': outcomes,
                    f's{S}_outcome_count': outcome_count, f's{S}_traj_boards': t_boards, f's{S}_traj_actions': t_actions,
                    f's{S}_traj_reward': t_reward, f's{S}_traj_done': t_done,
                    f's{S}_traj_meta': np.array([tseed, env_base, step0], dtype=np.int64)})
        total += n
    np.savez_compressed(OUT / 'nxn.npz', **out)
    return total, int(sum(out[f's{S}_traj_done'].sum() for S in NXN_SIZES))


def gen_random_play_stats(games: int = 3000):
    """Whole games of the reference engine under uniformly random actions with ITS OWN spawn RNG (seeded per step):
    the bitstream cannot be reproduced, the DISTRIBUTION of game length / score / max tile can -- it pins
    P(tile 4) = 0.1, the uniform choice of the spawn cell and the invalid-move handling end to end."""
    rng = np.random.default_rng(20482)
    length = np.zeros(games, dtype=np.int32)
    total = np.zeros(games, dtype=np.int64)
    max_tile = np.zeros(games, dtype=np.int32)
    invalid = np.zeros(games, dtype=np.int32)
    seed = 5_000_000
    for g in range(games):
        state = np.zeros((4, 4), dtype=np.int64)
        state = ref_board.fill_cells(state=state, number_tile=2, seed=seed)
        seed += 1
        steps = 0
        while not ref_board.is_done(state):
            new_state, reward = ref_board.next_state(state, int(rng.integers(0, 4)), seed=seed)
            seed += 1
            steps += 1
            invalid[g] += int(new_state is state)
            total[g] += int(reward)
            state = new_state
        length[g] = steps
        max_tile[g] = int(state.max())
    np.savez_compressed(OUT / 'random_play_stats.npz', length=length, total_reward=total, max_tile=max_tile, invalid_moves=invalid)
    return games, float(length.mean()), float(total.mean()), float(invalid.sum() / length.sum())


def write_checksums():
    with open(OUT / 'SHA256SUMS', 'w') as fh:
        for p in sorted(OUT.glob('*.npz')):
            fh.write(f'{hashlib.sha256(p.read_bytes()).hexdigest()}  {p.name}\n')
            print(p.name, p.stat().st_size)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[1] == '--only':  # regenerate ONE fixture set (each is a deterministic function of its inputs)
        print(sys.argv[2], globals()['gen_' + sys.argv[2]]())
        write_checksums()
        return
    rng = np.random.default_rng(20480)
    print('rows', gen_rows())
    print('afterstates', gen_afterstates(rng))
    print('next_state', gen_next_state(rng))
    print('trajectories', gen_trajectories())
    print('resets', gen_resets())
    print('transforms', gen_transforms(rng))
    print('chance_outcomes', gen_chance_outcomes())
    print('expectimax_d2', gen_expectimax_d2())
    print('expectimax_d3', gen_expectimax_d3())
    print('expectimax_d4', gen_expectimax_d4())
    print('random_play_stats', gen_random_play_stats())
    print('nxn', gen_nxn())
    write_checksums()


if __name__ == '__main__':
    main()
