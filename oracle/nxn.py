"""
Python face of the oracle's S x S functions (2 <= S <= 8) -- TEST INFRASTRUCTURE like the rest of ``oracle``.
Boards are uint8 exponent arrays of shape (n, S * S), row-major.  [ref: the reference's core functions take boards of any
shape, twentyfortyeight/core/gameboard.py / gamemove.py; `size` of twentyfortyeight/envs/twentyfortyeight.py:21]
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _ptr, _u8p, lib

_declared = False


def _lib():
    global _declared
    L = lib()
    if not _declared:
        u8p, i32p, u64, i = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32), ctypes.c_uint64, ctypes.c_int
        L.oracle_nxn_merge_column.argtypes = [u8p, i, u8p, ctypes.POINTER(ctypes.c_int)]
        L.oracle_nxn_merge_column.restype = ctypes.c_int64
        L.oracle_nxn_step_batch.argtypes = [u64, i, u64, u8p, u8p, i32p, u8p, u8p, u64, u64, ctypes.c_uint32]
        L.oracle_nxn_afterstate_batch.argtypes = [u64, i, u8p, u8p, u8p, i32p, u8p]
        L.oracle_nxn_status_batch.argtypes = [u64, i, u8p, u8p, u8p, u8p, u8p]
        L.oracle_nxn_reset_batch.argtypes = [u64, i, u64, u8p, u64, u64]
        L.oracle_nxn_spawn_random_batch.argtypes = [u64, i, u64, u8p, u64, u64, ctypes.c_uint32]
        L.oracle_nxn_chance_outcomes.argtypes = [u64, i, u8p, u8p, u8p]
        for name in ('step_batch', 'afterstate_batch', 'status_batch', 'reset_batch', 'spawn_random_batch', 'chance_outcomes'):
            getattr(L, 'oracle_nxn_' + name).restype = ctypes.c_int
        _declared = True
    return L


def _boards(a, size):
    b = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, size * size)
    return b


def merge_column(column):
    c = np.ascontiguousarray(column, dtype=np.uint8)
    out = np.zeros(8, dtype=np.uint8)
    n = ctypes.c_int()
    score = _lib().oracle_nxn_merge_column(_u8p(c), c.shape[0], _u8p(out), ctypes.byref(n))
    return int(score), out[:n.value].copy()


def step_batch(boards, size, actions, seed, step_index, *, env_base=0, flags=0, want_legal=False):
    b = _boards(boards, size).copy()
    n = b.shape[0]
    a = None if actions is None else np.ascontiguousarray(actions, dtype=np.uint8)
    reward = np.zeros(n, dtype=np.int32)
    done = np.zeros(n, dtype=np.uint8)
    legal = np.zeros(n, dtype=np.uint8) if want_legal else None
    rc = _lib().oracle_nxn_step_batch(n, size, env_base, _u8p(b), _u8p(a), _ptr(reward, ctypes.c_int32), _u8p(done), _u8p(legal), seed,
                                      step_index, flags | (2 if a is None else 0))
    assert rc == 0
    out = {'boards': b, 'reward': reward, 'done': done}
    if want_legal:
        out['legal'] = legal
    return out


def afterstate_batch(boards, size, actions):
    b = _boards(boards, size)
    n = b.shape[0]
    a = np.ascontiguousarray(actions, dtype=np.uint8)
    out = np.zeros_like(b)
    reward = np.zeros(n, dtype=np.int32)
    changed = np.zeros(n, dtype=np.uint8)
    assert _lib().oracle_nxn_afterstate_batch(n, size, _u8p(b), _u8p(a), _u8p(out), _ptr(reward, ctypes.c_int32), _u8p(changed)) == 0
    return {'boards': out, 'reward': reward, 'changed': changed}


def status_batch(boards, size):
    b = _boards(boards, size)
    n = b.shape[0]
    done, legal, n_empty, max_exp = (np.zeros(n, dtype=np.uint8) for _ in range(4))
    assert _lib().oracle_nxn_status_batch(n, size, _u8p(b), _u8p(done), _u8p(legal), _u8p(n_empty), _u8p(max_exp)) == 0
    return {'done': done, 'legal': legal, 'n_empty': n_empty, 'max_exp': max_exp}


def reset_batch(n, size, seed, step_index, *, env_base=0):
    b = np.zeros((n, size * size), dtype=np.uint8)
    assert _lib().oracle_nxn_reset_batch(n, size, env_base, _u8p(b), seed, step_index) == 0
    return b


def spawn_random_batch(boards, size, seed, step_index, count, *, env_base=0):
    b = _boards(boards, size).copy()
    assert _lib().oracle_nxn_spawn_random_batch(b.shape[0], size, env_base, _u8p(b), seed, step_index, count) == 0
    return b


def chance_outcomes(boards, size):
    b = _boards(boards, size)
    n, cells = b.shape[0], size * size
    out = np.zeros((n, 2 * cells, cells), dtype=np.uint8)
    count = np.zeros(n, dtype=np.uint8)
    assert _lib().oracle_nxn_chance_outcomes(n, size, _u8p(b), _u8p(out), _u8p(count)) == 0
    return out, count
