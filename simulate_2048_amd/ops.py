"""
Device operations on packed boards -- thin, allocation-doing wrappers around the C-ABI, used by the
API mirrors.  ``boards`` are uint8 tensors of shape (n, 16) on the runtime's device; every function
is asynchronous on torch's current stream (reading a result back synchronises as usual).
"""

from __future__ import annotations

import threading

import numpy as np
import torch

from . import capi, runtime


def _p(t):
    return None if t is None else t.data_ptr()


def _u8(x, n, rt) -> torch.Tensor:
    t = torch.as_tensor(x)
    if t.dtype != torch.uint8:
        t = (t.to(torch.int64) & 0xFF).to(torch.uint8)
    t = t.reshape(n).to(rt.device).contiguous()
    return t


def pack(values, rt=None) -> torch.Tensor:
    """Tile VALUES (array-like of shape (..., 4, 4) or (..., 16), any integer dtype) -> packed (n,16) uint8."""
    rt = rt or runtime.current()
    v = torch.as_tensor(np.asarray(values)) if not isinstance(values, torch.Tensor) else values
    if v.dtype not in (torch.int32, torch.int64):
        v = v.to(torch.int64)
    v = v.reshape(-1, 16).to(rt.device).contiguous()
    n = v.shape[0]
    out = torch.empty((n, 16), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    (rt.lib.pack_i32 if v.dtype == torch.int32 else rt.lib.pack_i64)(dev, stream, n, _p(v), _p(out))
    return out


def unpack(boards: torch.Tensor, dtype=torch.int64, rt=None) -> torch.Tensor:
    """Packed (n,16) uint8 -> tile values (n,16) of dtype int32 or int64."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = torch.empty((n, 16), dtype=dtype, device=rt.device)
    dev, stream = rt.handles()
    (rt.lib.unpack_i32 if dtype == torch.int32 else rt.lib.unpack_i64)(dev, stream, n, _p(boards), _p(out))
    return out


def afterstate(boards: torch.Tensor, actions, rt=None):
    """-> (afterstate boards, reward int32, changed uint8); input untouched."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    a = _u8(actions, n, rt)
    out = torch.empty_like(boards)
    reward = torch.empty((n,), dtype=torch.int32, device=rt.device)
    changed = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.afterstate(dev, stream, n, _p(boards), _p(a), _p(out), _p(reward), _p(changed))
    return out, reward, changed


def afterstates4(boards: torch.Tensor, rt=None):
    """Afterstates of all four actions: (boards (n,4,16) uint8, reward (n,4) int32, legal bitmask (n,) uint8)."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = torch.empty((n, 4, 16), dtype=torch.uint8, device=rt.device)
    reward = torch.empty((n, 4), dtype=torch.int32, device=rt.device)
    legal = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.afterstates4(dev, stream, n, _p(boards), _p(out), _p(reward), _p(legal))
    return out, reward, legal


def spawn(boards: torch.Tensor, spawn_k, spawn_e, rt=None) -> torch.Tensor:
    """In place: exponent spawn_e[i] into the spawn_k[i]-th empty cell. Returns placed (uint8, 255 = none)."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    k, e = _u8(spawn_k, n, rt), _u8(spawn_e, n, rt)
    placed = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.spawn(dev, stream, n, _p(boards), _p(k), _p(e), _p(placed))
    return placed


def spawn_random(boards: torch.Tensor, seed: int, step_index: int, count: int = 1, env_base: int = 0, rt=None) -> None:
    rt = rt or runtime.current()
    dev, stream = rt.handles()
    rt.lib.spawn_random(dev, stream, boards.shape[0], env_base, _p(boards), seed, step_index, count)


def status(boards: torch.Tensor, *, done=True, legal=True, obs=False, max_exp=False, n_empty=False, rt=None) -> dict:
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = {}
    if done:
        out['done'] = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    if legal:
        out['legal'] = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    if obs:
        out['obs'] = torch.empty((n, 16), dtype=torch.float32, device=rt.device)
    if max_exp:
        out['max_exp'] = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    if n_empty:
        out['n_empty'] = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.status(dev, stream, n, _p(boards), _p(out.get('done')), _p(out.get('legal')), _p(out.get('obs')),
                  _p(out.get('max_exp')), _p(out.get('n_empty')))
    return out


def chance_outcomes(boards: torch.Tensor, rt=None):
    """All spawn outcomes of every afterstate: (outcome boards (n,32,16) uint8, prob (n,32) f32, count (n,) uint8)."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = torch.empty((n, 32, 16), dtype=torch.uint8, device=rt.device)
    prob = torch.empty((n, 32), dtype=torch.float32, device=rt.device)
    count = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.chance_outcomes(dev, stream, n, _p(boards), _p(out), _p(prob), _p(count))
    return out, prob, count


def expectimax_ply(boards: torch.Tensor, *, want_status: bool = True, rt=None) -> dict:
    """One fused expectimax ply: every board's 4 x 32 successor slots (env2048_expectimax_ply).
    -> dict(boards (n,4,32,16) u8, prob (n,4,32) f32, count (n,4) u8, reward (n,4) i32, legal (n,) u8[, status (n,4,32) u8])."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = {
        'boards': torch.empty((n, 4, 32, 16), dtype=torch.uint8, device=rt.device),
        'prob': torch.empty((n, 4, 32), dtype=torch.float32, device=rt.device),
        'count': torch.empty((n, 4), dtype=torch.uint8, device=rt.device),
        'reward': torch.empty((n, 4), dtype=torch.int32, device=rt.device),
        'legal': torch.empty((n,), dtype=torch.uint8, device=rt.device),
    }
    if want_status:
        out['status'] = torch.empty((n, 4, 32), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.expectimax_ply(dev, stream, n, _p(boards), _p(out['boards']), _p(out['prob']), _p(out.get('status')), _p(out['count']),
                          _p(out['reward']), _p(out['legal']))
    return out


def expectimax_depth2(boards: torch.Tensor, *, want_terms: bool = False, rt=None) -> dict:
    """Depth-2 expectimax of the true environment for every board in ONE launch (env2048_expectimax_depth2): Q2(s, a) = r(s, a) +
    E over the spawn outcomes of the afterstate of the best immediate reward (the two-ply tree is evaluated on chip).
    -> dict(q (n,4) f32, -inf for illegal actions; best (n,) u8, 255 = terminal; value (n,) f32[, terms (n,4,4) i32 = {r, S2, S4, num_empty}])."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = {
        'q': torch.empty((n, 4), dtype=torch.float32, device=rt.device),
        'best': torch.empty((n,), dtype=torch.uint8, device=rt.device),
        'value': torch.empty((n,), dtype=torch.float32, device=rt.device),
    }
    if want_terms:
        out['terms'] = torch.empty((n, 4, 4), dtype=torch.int32, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.expectimax_depth2(dev, stream, n, _p(boards), _p(out.get('terms')), _p(out['q']), _p(out['best']), _p(out['value']))
    return out


_slots = threading.local()


def _total_slot(rt, stream):
    """The calling thread's two-word host slot for env2048_expand_offsets on `stream` (total, completion word) -- pinned on a HIP
    runtime -- and the next completion value to use with it.  One slot per (device, stream): calls on one stream are ordered,
    so a slot is never written by two launches whose order is unknown."""
    table = getattr(_slots, 'table', None)
    if table is None:
        table = _slots.table = {}
    key = (rt.device, stream)
    entry = table.get(key)
    if entry is None:
        # (never freed: a launch may still be writing a slot, and 16 bytes per stream a thread ever used is nothing)
        slot = torch.zeros((2,), dtype=torch.int64)
        if rt.device.type == 'cuda':
            slot = slot.pin_memory()
        entry = table[key] = [rt.device, slot, slot.numpy(), slot.data_ptr(), 0]
    entry[4] += 1
    return entry


def expand(boards: torch.Tensor, rt=None) -> dict:
    """One level of the search tree, densely: the children of every board (the (empty cell, tile) outcomes of its legal
    afterstates, in the order action / cell rank / tile 2, 4) as the next frontier.  -> dict(children (total, 16) u8,
    offsets (n,) int64 = first child of every board, reward (n, 4) i32, n_empty (n, 4) u8 (0 = illegal action)).
    env2048_expand_offsets (count + scan on the device; the total arrives in pinned host memory behind a completion word, so
    sizing the frontier costs a few microseconds of polling instead of a stream synchronisation and a copy) -> env2048_expand."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    if n == 0:
        return {'children': torch.empty((0, 16), dtype=torch.uint8, device=rt.device), 'offsets': torch.empty((0,), dtype=torch.int64, device=rt.device),
                'reward': torch.empty((0, 4), dtype=torch.int32, device=rt.device), 'n_empty': torch.empty((0, 4), dtype=torch.uint8, device=rt.device)}
    dev, stream = rt.handles()
    offsets = torch.empty((n,), dtype=torch.int64, device=rt.device)
    scratch = torch.empty((rt.lib.expand_scratch_words(n),), dtype=torch.int64, device=rt.device)
    _, slot, words, address, serial = _total_slot(rt, stream)
    rt.lib.expand_offsets(dev, stream, n, _p(boards), _p(offsets), _p(scratch), address, serial)
    out = {'offsets': offsets, 'reward': torch.empty((n, 4), dtype=torch.int32, device=rt.device),
           'n_empty': torch.empty((n, 4), dtype=torch.uint8, device=rt.device)}  # (allocated while the device counts)
    if not rt.lib.wait_sequence(address + 8, serial, 200):  # (a 2^20-board level counts for ~100 us; beyond the window: the general way)
        rt.synchronize()
        if int(words[1]) != serial:
            raise RuntimeError('env2048_expand_offsets: the completion word did not arrive')
    total = int(words[0])
    out['children'] = torch.empty((total, 16), dtype=torch.uint8, device=rt.device)
    rt.lib.expand(dev, stream, n, _p(boards), _p(offsets), _p(out['children']) if total else None, _p(out['reward']), _p(out['n_empty']))
    return out


def expectimax(boards: torch.Tensor, depth: int, rt=None) -> dict:
    """Depth-d expectimax of the true environment (d >= 2) with the tree kept on the device: d - 2 expansions into dense
    frontiers (about 34 children per mid-game board and level), the depth-2 kernel on the deepest frontier, d - 2 back-ups.
    -> dict(q (n,4) f64, -inf for illegal actions; best (n,) u8, 255 = terminal; value (n,) f64, 0 for a terminal board;
    frontier = boards per level).  Memory: 16 B per frontier board -- 2^12 roots at depth 4 are about 5 million leaves."""
    if depth < 2:
        raise ValueError('expectimax depth must be >= 2')
    rt = rt or runtime.current()
    dev, stream = rt.handles()
    levels, frontier, sizes = [], boards, [boards.shape[0]]
    for _ in range(depth - 2):
        level = expand(frontier, rt=rt)
        levels.append((frontier.shape[0], level))
        frontier = level['children']
        sizes.append(frontier.shape[0])
    m = frontier.shape[0]
    value = torch.empty((m,), dtype=torch.float64, device=rt.device)
    if depth == 2:
        terms = torch.empty((m, 4, 4), dtype=torch.int32, device=rt.device)
        best = torch.empty((m,), dtype=torch.uint8, device=rt.device)
        rt.lib.expectimax_depth2(dev, stream, m, _p(frontier), _p(terms), None, _p(best), None)
        t = terms.to(torch.int64)
        d = 10 * t[..., 3]
        q = torch.where(t[..., 3] > 0, (d * t[..., 0] + 9 * t[..., 1] + t[..., 2]).double() / d.clamp(min=1).double(),
                        torch.full((), float('-inf'), dtype=torch.float64, device=rt.device))
        value = torch.where(best != 255, q.max(dim=1).values, torch.zeros((), dtype=torch.float64, device=rt.device))
        return {'q': q, 'best': best, 'value': value, 'frontier': sizes}
    if m:
        rt.lib.expectimax_depth2_f64(dev, stream, m, _p(frontier), _p(value))
    q = best = None
    for k, (n_level, level) in reversed(list(enumerate(levels))):
        parent = torch.empty((n_level,), dtype=torch.float64, device=rt.device)
        if k == 0:
            q = torch.empty((n_level, 4), dtype=torch.float64, device=rt.device)
            best = torch.empty((n_level,), dtype=torch.uint8, device=rt.device)
        if n_level:
            rt.lib.expectimax_backup(dev, stream, n_level, _p(level['reward']), _p(level['n_empty']), _p(level['offsets']),
                                     _p(value) if value.numel() else None, _p(parent), _p(q) if k == 0 else None, _p(best) if k == 0 else None)
        value = parent
    return {'q': q, 'best': best, 'value': value, 'frontier': sizes}


def onehot(boards: torch.Tensor, width: int = 31, rt=None) -> torch.Tensor:
    rt = rt or runtime.current()
    n = boards.shape[0]
    out = torch.empty((n, 16 * width), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.encode_onehot(dev, stream, n, _p(boards), width, _p(out))
    return out


def reset(n: int, seed: int, step_index: int = 0, env_base: int = 0, rt=None) -> torch.Tensor:
    rt = rt or runtime.current()
    boards = torch.empty((n, 16), dtype=torch.uint8, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.reset(dev, stream, n, env_base, _p(boards), None, seed, step_index)
    return boards


def step(boards: torch.Tensor, actions, seed: int, step_index: int, *, env_base: int = 0, flags: int = 0,
         want_legal: bool = False, want_obs: bool = False, rt=None) -> dict:
    """In place on ``boards``. Returns dict(reward, done[, legal, obs])."""
    rt = rt or runtime.current()
    n = boards.shape[0]
    a = None if actions is None else _u8(actions, n, rt)
    if a is None:
        flags |= capi.FLAG_RANDOM_ACTIONS
    out = {
        'reward': torch.empty((n,), dtype=torch.int32, device=rt.device),
        'done': torch.empty((n,), dtype=torch.uint8, device=rt.device),
    }
    if want_legal:
        out['legal'] = torch.empty((n,), dtype=torch.uint8, device=rt.device)
    if want_obs:
        out['obs'] = torch.empty((n, 16), dtype=torch.float32, device=rt.device)
    dev, stream = rt.handles()
    rt.lib.step(dev, stream, n, env_base, _p(boards), _p(a), _p(out['reward']), _p(out['done']), _p(out.get('legal')),
                _p(out.get('obs')), seed, step_index, flags)
    return out
