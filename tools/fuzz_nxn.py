#!/usr/bin/env python3
"""Randomised differential run of the general-size kernels (env2048_nxn_*) against the oracle's S x S restatement (GPU box):
random sizes 2 ... 8, batch sizes, index offsets, seeds, board mixtures (random play, sparse, dense, big exponents), all six
entry points.  python tools/fuzz_nxn.py [seconds]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from oracle import nxn as ref
from simulate_2048_amd import capi

lib = capi.library()
rng = np.random.default_rng(int(time.time()))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
counts = {'step': 0, 'afterstate': 0, 'status': 0, 'spawn': 0, 'outcomes': 0, 'reset': 0, 'board_steps': 0}


def boards_for(n, cells):
    kind = rng.integers(0, 4)
    if kind == 0:
        b = rng.integers(0, 6, size=(n, cells))
    elif kind == 1:
        b = np.where(rng.random((n, cells)) < 0.06, 0, rng.integers(1, 4, size=(n, cells)))
    elif kind == 2:
        b = np.where(rng.random((n, cells)) < 0.3, 0, rng.integers(0, 28, size=(n, cells)))
    else:
        b = np.where(rng.random((n, cells)) < 0.85, 0, rng.integers(1, 3, size=(n, cells)))
    return b.astype(np.uint8)


t0 = last = time.time()
while time.time() - t0 < budget:
    size = int(rng.integers(2, 9))
    cells = size * size
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 1000, int(rng.integers(1, 6000))]))
    base = int(rng.choice([0, 1, 3, int(rng.integers(0, 1 << 40))]))
    seed = int(rng.integers(0, 1 << 62))
    step = int(rng.integers(0, 1 << 40))
    host = boards_for(n, cells)
    if rng.random() < 0.3:  # boards from play: reset + a few random steps
        host = ref.reset_batch(n, size, seed, step, env_base=base)
        dev = torch.zeros((n, cells), dtype=torch.uint8, device='cuda')
        lib.nxn_reset(0, S(), n, size, base, P(dev), None, seed, step)
        assert np.array_equal(dev.cpu().numpy(), host)
        counts['reset'] += 1
    shift = int(rng.choice([0, 0, 1, 4, 8, int(rng.integers(0, 16))]))  # the batch at any byte offset (unaligned: the per-lane byte path)
    arena = torch.full((shift + n * cells + 16,), 0xEE, dtype=torch.uint8, device='cuda')
    dev = arena[shift:shift + n * cells].view(n, cells)
    dev.copy_(cu(host))
    reward = torch.full((n,), -3, dtype=torch.int32, device='cuda')
    done = torch.full((n,), 7, dtype=torch.uint8, device='cuda')
    legal = torch.full((n,), 99, dtype=torch.uint8, device='cuda')
    for t in range(int(rng.integers(1, 12))):
        random_actions = rng.random() < 0.3
        acts = None if random_actions else rng.integers(0, 256, size=n).astype(np.uint8)
        flags = int(rng.integers(0, 2)) | (capi.FLAG_RANDOM_ACTIONS if random_actions else 0)
        lib.nxn_step(0, S(), n, size, base, P(dev), None if random_actions else P(cu(acts)), P(reward), P(done), P(legal), seed, step + t, flags)
        want = ref.step_batch(host, size, None if random_actions else acts & 3, seed, step + t, env_base=base, flags=flags & 1, want_legal=True)
        host = want['boards']
        assert np.array_equal(dev.cpu().numpy(), host), (size, n, t)
        assert np.array_equal(reward.cpu().numpy(), want['reward']) and np.array_equal(done.cpu().numpy(), want['done'])
        assert np.array_equal(legal.cpu().numpy(), want['legal'])
        counts['step'] += 1
        counts['board_steps'] += n
    assert bool((arena[:shift] == 0xEE).all()) and bool((arena[shift + n * cells:] == 0xEE).all()), 'bytes outside the batch were written'
    acts = rng.integers(0, 4, size=n).astype(np.uint8)
    out = torch.zeros_like(dev)
    changed = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.nxn_afterstate(0, S(), n, size, P(dev), P(cu(acts)), P(out), P(reward), P(changed))
    want = ref.afterstate_batch(host, size, acts)
    assert np.array_equal(out.cpu().numpy(), want['boards']) and np.array_equal(reward.cpu().numpy(), want['reward'])
    assert np.array_equal(changed.cpu().numpy(), want['changed'])
    counts['afterstate'] += 1
    st = {k: torch.zeros((n,), dtype=torch.uint8, device='cuda') for k in ('done', 'legal', 'n_empty', 'max_exp')}
    lib.nxn_status(0, S(), n, size, P(dev), P(st['done']), P(st['legal']), P(st['n_empty']), P(st['max_exp']))
    want = ref.status_batch(host, size)
    for k in st:
        assert np.array_equal(st[k].cpu().numpy(), want[k]), k
    counts['status'] += 1
    sp = dev.clone()
    c = int(rng.integers(1, 5))
    lib.nxn_spawn_random(0, S(), n, size, base, P(sp), seed, step + 100, c)
    assert np.array_equal(sp.cpu().numpy(), ref.spawn_random_batch(host, size, seed, step + 100, c, env_base=base))
    counts['spawn'] += 1
    m = min(n, 64)
    outcomes = torch.full((m, 2 * cells, cells), 9, dtype=torch.uint8, device='cuda')
    count = torch.zeros((m,), dtype=torch.uint8, device='cuda')
    lib.nxn_chance_outcomes(0, S(), m, size, P(dev), P(outcomes), P(count))
    want_o, want_c = ref.chance_outcomes(host[:m], size)
    assert np.array_equal(outcomes.cpu().numpy(), want_o) and np.array_equal(count.cpu().numpy(), want_c)
    counts['outcomes'] += 1
    if time.time() - last > 60:
        last = time.time()
        print('fuzz_nxn:', counts, flush=True)
torch.cuda.synchronize()
print('fuzz_nxn done, all identical:', counts)
