"""
Device operations on packed boards -- thin, allocation-doing wrappers around the C-ABI, used by the
API mirrors.  ``boards`` are uint8 tensors of shape (n, 16) on the runtime's device; every function
is asynchronous on torch's current stream (reading a result back synchronises as usual).
"""

from __future__ import annotations

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
