"""
CPU unit tests of the HIP kernels' per-board arithmetic: simulate_2048_amd/csrc/board_ops.hpp is
compiled for the host (ENV2048_HOST_EMULATION: v_perm_b32 / popcount / mulhi emulated), compared bit-for-bit with
the oracle and the golden fixtures as a shared object built with -fsanitize=undefined (UBSan), and run once more as a
standalone executable under -fsanitize=address,undefined (tests/host_emu/asan_main.cpp: the batch loops over caller
buffers on exactly-sized heap blocks).  Sanitizers run on this CPU build only (none are available for the GPU here).
This is a test harness (tests/host_emu), not a CPU backend: the product never loads it.
"""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle
from conftest import REPO

SRC = REPO / 'tests' / 'host_emu' / 'emu.cpp'
HDR = REPO / 'simulate_2048_amd' / 'csrc' / 'board_ops.hpp'
OUT = REPO / 'tests' / 'host_emu' / '_build' / 'libemu.so'


def u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


@pytest.fixture(scope='module')
def emu():
    OUT.parent.mkdir(exist_ok=True)
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, HDR.stat().st_mtime):
        cmd = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-fsanitize=undefined', '-fno-sanitize-recover=all',
               '-Wall', '-Wextra', f'-I{HDR.parent}', str(SRC), '-o', str(OUT)]
        subprocess.run(cmd, check=True)
    return ctypes.CDLL(str(OUT))


def emu_afterstate(emu, boards, actions, mode):
    n = boards.shape[0]
    b = np.ascontiguousarray(boards, dtype=np.uint8)
    a = np.ascontiguousarray(actions, dtype=np.uint8)
    out = np.zeros_like(b)
    reward = np.zeros(n, dtype=np.int32)
    changed = np.zeros(n, dtype=np.uint8)
    emu.emu_afterstate(ctypes.c_uint64(n), u8(b), u8(a), u8(out), ptr(reward, ctypes.c_int32), u8(changed), mode)
    return out, reward, changed


def test_philox_kat(emu):
    out = np.zeros(4, dtype=np.uint32)
    ctr = np.array([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], dtype=np.uint32)
    key = np.array([0xA4093822, 0x299F31D0], dtype=np.uint32)
    emu.emu_philox(ptr(ctr, ctypes.c_uint32), ptr(key, ctypes.c_uint32), ptr(out, ctypes.c_uint32))
    assert out.tolist() == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


@pytest.mark.parametrize('mode', [0, 1])
def test_afterstate_golden(emu, golden, mode):
    g = golden['afterstates']
    n = g['boards'].shape[0]
    for a in range(4):
        out, reward, changed = emu_afterstate(emu, g['boards'], np.full(n, a, dtype=np.uint8), mode)
        np.testing.assert_array_equal(out, g['after'][:, a])
        np.testing.assert_array_equal(reward, g['reward'][:, a])
        np.testing.assert_array_equal(changed, g['valid'][:, a])


@pytest.mark.parametrize('mode', [0, 1])
def test_rows_exhaustive(emu, golden, mode):
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
            boards[:, 3, :] = line
        else:
            boards[:, :, 2] = line
        out, reward, changed = emu_afterstate(emu, boards.reshape(n, 16), np.full(n, action, dtype=np.uint8), mode)
        got = out.reshape(n, 4, 4)
        np.testing.assert_array_equal(got[:, 3, :] if action in (0, 2) else got[:, :, 2], want)
        np.testing.assert_array_equal(reward, g['score'])
        np.testing.assert_array_equal(changed, g['can_move'])


def test_line_scores_exhaustive(emu, golden):
    """line_merge_terms / line_score (the per-line scores the expectimax leaf evaluator is built on): every one of the 18^4
    exponent rows scores what the reference's merge_column scored it (rows.npz), as any of the four lines of a group; the sum
    over a group equals slide_merge_planes' score; exponents up to 30 (merging into 2^31) are covered by random lines against the same slide."""
    g = golden['rows']
    base = int(g['max_e']) + 1
    idx = np.arange(base**4)
    rows = np.stack([(idx // base**3) % base, (idx // base**2) % base, (idx // base) % base, idx % base], axis=1).astype(np.uint8)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 31, size=(40_000, 4)).astype(np.uint8) * (rng.random((40_000, 4)) < 0.7)  # (a merge of two 2^31 tiles is outside the domain)
    eq = rng.random(40_000) < 0.6
    big[eq, 1] = big[eq, 0]
    eq2 = rng.random(40_000) < 0.4
    big[eq2, 3] = big[eq2, 2]
    for lines, want in ((rows, g['score'].astype(np.uint32)), (rows[rng.permutation(rows.shape[0])], None), (big.astype(np.uint8), None)):
        lines = np.ascontiguousarray(lines)
        n = lines.shape[0]
        scores = np.zeros(n, dtype=np.uint32)
        sums = np.zeros((n + 3) // 4, dtype=np.uint32)
        slides = np.zeros((n + 3) // 4, dtype=np.uint32)
        emu.emu_line_scores(ctypes.c_uint64(n), u8(lines), ptr(scores, ctypes.c_uint32), ptr(sums, ctypes.c_uint32), ptr(slides, ctypes.c_uint32))
        if want is not None:
            np.testing.assert_array_equal(scores, want)
        np.testing.assert_array_equal(sums, slides)
        padded = np.concatenate([scores, np.zeros((-n) % 4, dtype=np.uint32)]).reshape(-1, 4)
        np.testing.assert_array_equal(padded.sum(axis=1, dtype=np.uint32), sums)


def test_status_and_spawn(emu, golden):
    g = golden['afterstates']
    boards = np.ascontiguousarray(g['boards'])
    n = boards.shape[0]
    done, legal, n_empty = (np.zeros(n, dtype=np.uint8) for _ in range(3))
    emu.emu_status(ctypes.c_uint64(n), u8(boards), u8(done), u8(legal), u8(n_empty))
    np.testing.assert_array_equal(done, g['done'])
    np.testing.assert_array_equal(legal, g['legal'])
    np.testing.assert_array_equal(n_empty, (boards == 0).sum(axis=1))
    rng = np.random.default_rng(0)
    k = rng.integers(0, 17, size=n).astype(np.uint8)
    e = rng.integers(1, 3, size=n).astype(np.uint8)
    ref = oracle.spawn_batch(boards, k, e)
    b = boards.copy()
    placed = np.zeros(n, dtype=np.uint8)
    emu.emu_spawn(ctypes.c_uint64(n), u8(b), u8(k), u8(e), u8(placed))
    np.testing.assert_array_equal(b, ref['boards'])
    np.testing.assert_array_equal(placed, ref['placed'])


def occupancy_boards(rng, lo=1, hi=31):
    """All 65 536 occupancy patterns of a board; occupied cells hold random exponents in [lo, hi]."""
    occ = (np.arange(1 << 16, dtype=np.uint32)[:, None] >> np.arange(16, dtype=np.uint32)) & 1
    return (occ * rng.integers(lo, hi + 1, size=occ.shape)).astype(np.uint8)


def test_spawn_every_occupancy_every_rank(emu):
    """k-th empty cell, every occupancy pattern x every rank 0..16 (16 = always out of range), tiles 2 and 4,
    neighbours up to exponent 31 (the occupancy flags rely on value + 0x1F staying below bit 6)."""
    rng = np.random.default_rng(11)
    base = occupancy_boards(rng)
    for k in range(17):
        kk = np.full(base.shape[0], k, dtype=np.uint8)
        e = rng.integers(1, 3, size=base.shape[0]).astype(np.uint8)
        ref = oracle.spawn_batch(base, kk, e)
        b = base.copy()
        placed = np.zeros(base.shape[0], dtype=np.uint8)
        emu.emu_spawn(ctypes.c_uint64(b.shape[0]), u8(b), u8(kk), u8(e), u8(placed))
        np.testing.assert_array_equal(b, ref['boards'], err_msg=f'k={k}')
        np.testing.assert_array_equal(placed, ref['placed'], err_msg=f'k={k}')


def test_status_every_occupancy(emu):
    """done / legal / empty count on every occupancy pattern, with two tile values (so that equal and
    different neighbours both occur) and with all-different large exponents."""
    rng = np.random.default_rng(12)
    for lo, hi in ((1, 2), (30, 31), (1, 31)):
        boards = occupancy_boards(rng, lo, hi)
        n = boards.shape[0]
        ref = oracle.status_batch(boards)
        done, legal, n_empty = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        emu.emu_status(ctypes.c_uint64(n), u8(boards), u8(done), u8(legal), u8(n_empty))
        np.testing.assert_array_equal(done, ref['done'])
        np.testing.assert_array_equal(legal, ref['legal'])
        np.testing.assert_array_equal(n_empty, ref['n_empty'])


def test_step_words_match_oracle(emu):
    seed, step = 0x1122334455667788, (1 << 33) + 5
    for env_base in (0, 1001, (1 << 34) + 3):
        n = 777
        words = np.zeros(n, dtype=np.uint32)
        emu.emu_step_words(ctypes.c_uint64(n), ctypes.c_uint64(env_base), ctypes.c_uint64(seed), ctypes.c_uint64(step),
                           ptr(words, ctypes.c_uint32))
        assert words.tolist() == [oracle.step_word(seed, env_base + i, step) for i in range(n)]


@pytest.mark.parametrize('name', ['auto', 'plain'])
def test_trajectories_golden(emu, golden, name):
    g = golden['trajectories']
    seed, env_base, step0 = int(g[f'{name}_seed']), int(g[f'{name}_env_base']), int(g[f'{name}_step0'])
    flags = 1 if int(g[f'{name}_auto_reset']) else 0
    T, n = g[f'{name}_actions'].shape
    boards = np.zeros((n, 16), dtype=np.uint8)
    emu.emu_reset(ctypes.c_uint64(n), ctypes.c_uint64(env_base), u8(boards), ctypes.c_uint64(seed), ctypes.c_uint64(step0))
    np.testing.assert_array_equal(boards, g[f'{name}_boards0'])
    words = np.zeros(n, dtype=np.uint32)
    reward = np.zeros(n, dtype=np.int32)
    done, legal = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    for t in range(T):
        emu.emu_step_words(ctypes.c_uint64(n), ctypes.c_uint64(env_base), ctypes.c_uint64(seed), ctypes.c_uint64(step0 + t),
                           ptr(words, ctypes.c_uint32))
        acts = np.ascontiguousarray(g[f'{name}_actions'][t])
        emu.emu_step(ctypes.c_uint64(n), ctypes.c_uint64(env_base), u8(boards), u8(acts), ptr(words, ctypes.c_uint32),
                     ptr(reward, ctypes.c_int32), u8(done), u8(legal), ctypes.c_uint64(seed), ctypes.c_uint64(step0 + t), flags | 2)
        np.testing.assert_array_equal(boards, g[f'{name}_boards'][t], err_msg=f't={t}')
        np.testing.assert_array_equal(reward, g[f'{name}_reward'][t])
        np.testing.assert_array_equal(done, g[f'{name}_done'][t])
        np.testing.assert_array_equal(legal, g[f'{name}_legal'][t])


def test_random_boards_vs_oracle(emu):
    rng = np.random.default_rng(9)
    n = 40_000
    boards = np.where(rng.random((n, 16)) < 0.25, 0, rng.integers(0, 18, size=(n, 16))).astype(np.uint8)
    boards[: n // 2] = np.where(rng.random((n // 2, 16)) < 0.1, 0, rng.integers(1, 4, size=(n // 2, 16)))
    actions = rng.integers(0, 256, size=n).astype(np.uint8)
    seed, step, env_base = 5, 6, 7
    ref = oracle.step_batch(boards, actions, seed, step, env_base=env_base, flags=1, want_legal=True)
    words = np.zeros(n, dtype=np.uint32)
    emu.emu_step_words(ctypes.c_uint64(n), ctypes.c_uint64(env_base), ctypes.c_uint64(seed), ctypes.c_uint64(step),
                       ptr(words, ctypes.c_uint32))
    b = boards.copy()
    reward = np.zeros(n, dtype=np.int32)
    done, legal = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    emu.emu_step(ctypes.c_uint64(n), ctypes.c_uint64(env_base), u8(b), u8(actions), ptr(words, ctypes.c_uint32),
                 ptr(reward, ctypes.c_int32), u8(done), u8(legal), ctypes.c_uint64(seed), ctypes.c_uint64(step), 1)
    np.testing.assert_array_equal(b, ref['boards'])
    np.testing.assert_array_equal(reward, ref['reward'])
    np.testing.assert_array_equal(done, ref['done'])
    np.testing.assert_array_equal(legal, ref['legal'])


@pytest.mark.parametrize('flags', [0, 1, 2, 3])
def test_plane_space_transitions_equal_plain_steps(emu, flags):
    """Boards held in the plane space of their last move over T steps (transition_oriented: what the rollout / replay kernels
    run) and mapped back to rows once at the end == T plain transitions, for host actions and for actions taken from the step
    words, with and without auto-reset; start boards include empty, full and high-tile boards; every (previous action, action)
    pair occurs thousands of times.  The oracle checks the final boards and the sums as well."""
    rng = np.random.default_rng(40 + flags)
    n, T = 6000, 24
    seed, step0, env_base = 0xFEEDFACE12345678, (1 << 32) - 7, 5_000_000_003
    boards = np.where(rng.random((n, 16)) < 0.3, 0, rng.integers(0, 18, size=(n, 16))).astype(np.uint8)
    boards[: n // 2] = np.where(rng.random((n // 2, 16)) < 0.15, 0, rng.integers(1, 5, size=(n // 2, 16)))
    boards[0] = 0
    boards[1] = np.arange(1, 17)
    boards[2] = 17
    actions = rng.integers(0, 256, size=(T, n)).astype(np.uint8)
    words = np.zeros((T, n), dtype=np.uint32)
    for t in range(T):
        emu.emu_step_words(ctypes.c_uint64(n), ctypes.c_uint64(env_base), ctypes.c_uint64(seed), ctypes.c_uint64(step0 + t),
                           ptr(words[t], ctypes.c_uint32))
    plain = boards.copy()
    sums, dones = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.uint32)
    reward, done = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint8)
    for t in range(T):
        emu.emu_step(ctypes.c_uint64(n), ctypes.c_uint64(env_base), u8(plain), u8(actions[t]), ptr(words[t], ctypes.c_uint32),
                     ptr(reward, ctypes.c_int32), u8(done), None, ctypes.c_uint64(seed), ctypes.c_uint64(step0 + t), flags)
        sums += reward
        dones += done
    got = boards.copy()
    got_sums, got_dones, last_done = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint8)
    emu.emu_steps_oriented(ctypes.c_uint64(n), ctypes.c_uint64(env_base), u8(got), u8(actions), ptr(words, ctypes.c_uint32), ctypes.c_uint32(T),
                           ptr(got_sums, ctypes.c_int64), ptr(got_dones, ctypes.c_uint32), u8(last_done), ctypes.c_uint64(seed),
                           ctypes.c_uint64(step0), ctypes.c_uint32(flags))
    np.testing.assert_array_equal(got, plain)
    np.testing.assert_array_equal(got_sums, sums)
    np.testing.assert_array_equal(got_dones, dones)
    np.testing.assert_array_equal(last_done, done)
    want = oracle.rollout_batch(boards, seed, step0, T, env_base=env_base, flags=flags) if flags & 2 else None
    if want is not None:
        np.testing.assert_array_equal(got, want['boards'])
        np.testing.assert_array_equal(got_sums, want['return_sum'])


def test_orientation_tables_are_what_the_generator_derives():
    """ORIENT_SELECTORS / ORIENT_RANK_MASKS in board_ops.hpp == what tools/gen_orientation_tables.py derives from DIR_SELECTORS."""
    import re
    import subprocess
    import sys

    out = subprocess.run([sys.executable, str(REPO / 'tools' / 'gen_orientation_tables.py')], capture_output=True, text=True, check=True).stdout
    header = (REPO / 'simulate_2048_amd' / 'csrc' / 'board_ops.hpp').read_text()

    def values(text, start, stop):
        body = text[text.index(start):]
        body = body[:body.index(stop)]
        return [int(v, 16) for v in re.findall(r'0x([0-9A-Fa-f]{8})u', body)]

    gen_sel = values(out, 'constexpr uint32_t ORIENT_SELECTORS[64]', '};')
    gen_masks = values(out, 'constexpr uint32_t ORIENT_RANK_MASKS[16]', '};')
    assert len(gen_sel) == 64 and len(gen_masks) == 16
    assert values(header, '#define ENV2048_ORIENT_SELECTOR_VALUES', '#define ENV2048_ORIENT_RANK_MASK_VALUES') == gen_sel
    assert values(header, '#define ENV2048_ORIENT_RANK_MASK_VALUES', 'constexpr uint32_t ORIENT_SELECTORS') == gen_masks


def test_batch_loops_under_asan_and_ubsan(tmp_path):
    """ASan + UBSan over every batch function of the host build (the shared object only had UBSan)."""
    exe = tmp_path / 'emu_asan'
    main = REPO / 'tests' / 'host_emu' / 'asan_main.cpp'
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Wextra',
                    f'-I{HDR.parent}', f'-I{main.parent}', str(main), '-o', str(exe)], check=True)
    proc = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env={'ASAN_OPTIONS': 'detect_leaks=1'})
    assert proc.returncode == 0 and proc.stdout.startswith('asan+ubsan ok'), proc.stderr[-3000:]


def test_normalized_reward_and_max_exponent(emu, golden):
    """The device formula of ENV2048_FLAG_REWARD_LOG2_F32 (board_ops.hpp normalized_reward) against the reference's own
    normalize_reward values (tests/golden/transforms.npz, float64): tolerance 1 ulp of float32; and against the oracle for
    every score below 2^17 plus random ones below 2^24.  max_exponent against the oracle's board maximum."""
    g = golden['transforms']
    rewards = g['rewards'].astype(np.uint32)
    out = np.zeros(len(rewards), dtype=np.float32)
    emu.emu_normalized_reward(ctypes.c_uint64(len(rewards)), ptr(rewards, ctypes.c_uint32), ptr(out, ctypes.c_float))
    want = g['normalized'].astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(want), np.float32(1e-30)))
    assert np.all(np.abs(out - want) <= ulp), (out, want)
    assert out[0] == 0.0 and out[1] == 0.125
    rng = np.random.default_rng(0)
    scores = np.concatenate([np.arange(1 << 17, dtype=np.uint32), rng.integers(1, 1 << 24, size=200_000).astype(np.uint32)])
    got = np.zeros(len(scores), dtype=np.float32)
    emu.emu_normalized_reward(ctypes.c_uint64(len(scores)), ptr(scores, ctypes.c_uint32), ptr(got, ctypes.c_float))
    ref = np.where(scores == 0, 0.0, np.log2(np.maximum(scores, 1).astype(np.float64)) / 16.0).astype(np.float32)
    assert np.all(np.abs(got - ref) <= np.spacing(np.maximum(ref, np.float32(1e-30))))
    assert float(np.mean(got == ref)) > 0.9  # and almost always the correctly rounded value itself
    boards = rng.integers(0, 18, size=(5000, 16)).astype(np.uint8)
    mx = np.zeros(5000, dtype=np.uint8)
    emu.emu_max_exponent(ctypes.c_uint64(5000), u8(boards), u8(mx))
    np.testing.assert_array_equal(mx, boards.max(axis=1))
