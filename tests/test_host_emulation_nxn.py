"""
CPU unit tests of the general-size kernels' per-board arithmetic: simulate_2048_amd/csrc/nxn_ops.hpp (boards S x S, 2 <= S <= 8,
held in registers, four lines per dword) compiled for the host (ENV2048_HOST_EMULATION) with -fsanitize=undefined and compared
bit-for-bit with the oracle for every size: every line of short boards exhaustively, random boards of every density, whole
games with auto-reset.  A test harness (tests/host_emu), not a CPU backend: the product never loads it.
"""

import ctypes
import itertools
import subprocess

import numpy as np
import pytest

from conftest import REPO
from oracle import nxn as onxn

SRC = REPO / 'tests' / 'host_emu' / 'emu_nxn.cpp'
HDRS = [REPO / 'simulate_2048_amd' / 'csrc' / 'nxn_ops.hpp', REPO / 'simulate_2048_amd' / 'csrc' / 'board_ops.hpp']
OUT = REPO / 'tests' / 'host_emu' / '_build' / 'libemu_nxn.so'
SIZES = [2, 3, 4, 5, 6, 7, 8]
U64 = ctypes.c_uint64


def u8(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def i32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


@pytest.fixture(scope='module')
def emu():
    OUT.parent.mkdir(exist_ok=True)
    if not OUT.exists() or OUT.stat().st_mtime < max(p.stat().st_mtime for p in [SRC, *HDRS]):
        cmd = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-fsanitize=undefined', '-fno-sanitize-recover=all', '-Wall',
               '-Wextra', f'-I{HDRS[0].parent}', str(SRC), '-o', str(OUT)]
        subprocess.run(cmd, check=True)
    return ctypes.CDLL(str(OUT))


def random_boards(rng, n, size, top=12):
    """Boards of every density: per-board fill probability from empty to full, exponents 1 ... top (low tops merge a lot)."""
    cells = size * size
    fill = rng.random((n, 1))
    b = rng.integers(1, top + 1, (n, cells), dtype=np.uint8)
    b[rng.random((n, cells)) >= fill] = 0
    return b


def emu_afterstate(emu, boards, size, actions):
    n = boards.shape[0]
    out = np.zeros_like(boards)
    reward = np.zeros(n, dtype=np.int32)
    changed = np.zeros(n, dtype=np.uint8)
    assert emu.emu_nxn_afterstate(U64(n), size, u8(boards), u8(actions), u8(out), i32(reward), u8(changed)) == 0
    return out, reward, changed


@pytest.mark.parametrize('size', SIZES)
def test_every_line(emu, size):
    """Every line over {0, 1, 2, 3} (S <= 6) or a random 50 000 of them (S = 7, 8), as the first and the last row of otherwise empty boards (and as columns),
    slid in all four directions."""
    rng = np.random.default_rng(size)
    if size <= 6:
        lines = np.array(list(itertools.product(range(4), repeat=size)), dtype=np.uint8)
    else:
        lines = rng.integers(0, 4, (50_000, size), dtype=np.uint8)
    n = lines.shape[0]
    for where in (0, size - 1):
        boards = np.zeros((n, size, size), dtype=np.uint8)
        boards[:, where, :] = lines
        boards_t = np.ascontiguousarray(boards.transpose(0, 2, 1))
        for b in (boards, boards_t):
            flat = b.reshape(n, size * size)
            for action in range(4):
                acts = np.full(n, action, dtype=np.uint8)
                out, reward, changed = emu_afterstate(emu, flat, size, acts)
                ref = onxn.afterstate_batch(flat, size, acts)
                np.testing.assert_array_equal(out, ref['boards'])
                np.testing.assert_array_equal(reward, ref['reward'])
                np.testing.assert_array_equal(changed, ref['changed'])


@pytest.mark.parametrize('size', SIZES)
def test_random_boards_afterstate_status(emu, size):
    rng = np.random.default_rng(100 + size)
    for top in (2, 3, 12, 30):
        boards = random_boards(rng, 10_000, size, top)
        n = boards.shape[0]
        acts = rng.integers(0, 4, n, dtype=np.uint8)
        out, reward, changed = emu_afterstate(emu, boards, size, acts)
        ref = onxn.afterstate_batch(boards, size, acts)
        np.testing.assert_array_equal(out, ref['boards'])
        np.testing.assert_array_equal(reward, ref['reward'])
        np.testing.assert_array_equal(changed, ref['changed'])
        done, legal, n_empty, max_exp = (np.zeros(n, dtype=np.uint8) for _ in range(4))
        assert emu.emu_nxn_status(U64(n), size, u8(boards), u8(done), u8(legal), u8(n_empty), u8(max_exp)) == 0
        st = onxn.status_batch(boards, size)
        for name, got in (('done', done), ('legal', legal), ('n_empty', n_empty), ('max_exp', max_exp)):
            np.testing.assert_array_equal(got, st[name], err_msg=name)


@pytest.mark.parametrize('size', SIZES)
def test_place_every_rank(emu, size):
    """The k-th empty cell in row-major order, for every k of random occupancy patterns (and k past the end: nothing)."""
    rng = np.random.default_rng(200 + size)
    cells = size * size
    boards = random_boards(rng, 600, size)
    n = boards.shape[0]
    for k in range(cells + 1):
        b = boards.copy()
        kk = np.full(n, k, dtype=np.uint8)
        e = rng.integers(1, 3, n, dtype=np.uint8)
        assert emu.emu_nxn_place(U64(n), size, u8(b), u8(kk), u8(e)) == 0
        want = boards.copy()
        for i in range(n):
            empty = np.flatnonzero(want[i] == 0)
            if k < empty.shape[0]:
                want[i, empty[k]] = e[i]
        np.testing.assert_array_equal(b, want, err_msg=f'k={k}')


@pytest.mark.parametrize('size', SIZES)
def test_reset_and_games(emu, size):
    """Fresh boards and 200 steps of 1500 games with auto-reset (small boards end and restart many times), actions from the caller
    and from the step word, against the oracle every step."""
    seed, env_base = 0xFEEDFACE12345678, (1 << 33) + 5
    n = 1500
    boards = np.zeros((n, size * size), dtype=np.uint8)
    assert emu.emu_nxn_reset(U64(n), size, U64(env_base), U64(seed), U64(7), u8(boards)) == 0
    np.testing.assert_array_equal(boards, onxn.reset_batch(n, size, seed, 7, env_base=env_base))
    rng = np.random.default_rng(size)
    reward = np.zeros(n, dtype=np.int32)
    done, legal = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    ended = 0
    for t in range(200):
        given = t % 3 != 0
        acts = rng.integers(0, 4, n, dtype=np.uint8) if given else None
        flags = 1 | (0 if given else 2)
        ref = onxn.step_batch(boards, size, acts, seed, 100 + t, env_base=env_base, flags=1, want_legal=True)
        assert emu.emu_nxn_step(U64(n), size, U64(env_base), u8(boards), u8(acts), i32(reward), u8(done), u8(legal), U64(seed), U64(100 + t),
                                flags) == 0
        np.testing.assert_array_equal(boards, ref['boards'], err_msg=f't={t}')
        np.testing.assert_array_equal(reward, ref['reward'])
        np.testing.assert_array_equal(done, ref['done'])
        np.testing.assert_array_equal(legal, ref['legal'])
        ended += int(done.sum())
    if size <= 4:
        assert ended > 0
