#!/usr/bin/env python3
"""Randomised differential run of the kernels whose memory path changed in round 4 against the CPU oracle (GPU box):
env2048_afterstates4 (LDS-strip stores), env2048_status_i32 / env2048_afterstate_i32 / env2048_reset_keyed_i32_status
(coalesced int32 boards), env2048_chance_outcomes and env2048_encode_onehot (plain AND non-temporal stores: the choice is forced
both ways through child processes), env2048_count_episodes / env2048_segment_episodes (one and four games per lane, closed-form
count), env2048_random_actions (one lane per step block) -- random sizes incl. ragged waves, odd offsets, rings, frozen games,
float rewards.  python tools/fuzz_round4.py [seconds]"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if os.environ.get('FUZZ4_CHILD') is None:  # parent: one child per store policy (the policy is read once per process)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rc = 0
    for mode in ('', '0', '1'):
        env = dict(os.environ, FUZZ4_CHILD='1')
        if mode:
            env['ENV2048_EXP_STREAM_STORES'] = mode
        out = subprocess.run([sys.executable, __file__, str(budget / 3)], env=env, capture_output=True, text=True)
        print(f'store policy {"by size" if not mode else ("plain" if mode == "0" else "non-temporal")}:', out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
        rc |= out.returncode
    print('fuzz_round4 done, all identical' if rc == 0 else 'fuzz_round4 FAILED')
    sys.exit(rc)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from simulate_2048_amd import capi  # noqa: E402

lib = capi.library()
rng = np.random.default_rng(int(time.time()) ^ os.getpid())
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
SIZES = [1, 2, 3, 4, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 2049, 4100]
counts = {'afterstates4': 0, 'i32': 0, 'outcomes': 0, 'onehot': 0, 'episodes': 0, 'random_actions': 0}


def boards_for(n):
    kind = rng.integers(0, 3)
    if kind == 0:
        b = rng.integers(0, 6, size=(n, 16))
    elif kind == 1:
        b = np.where(rng.random((n, 16)) < 0.07, 0, rng.integers(1, 4, size=(n, 16)))
    else:
        b = np.where(rng.random((n, 16)) < 0.3, 0, rng.integers(0, 18, size=(n, 16)))
    return b.astype(np.uint8)


def pick_n():
    return int(rng.choice(SIZES + [int(rng.integers(1, 9000))]))


def fuzz_afterstates4():
    n = pick_n()
    b = boards_for(n)
    out = torch.full((n, 4, 16), 0x55, dtype=torch.uint8, device='cuda')
    reward = torch.zeros((n, 4), dtype=torch.int32, device='cuda')
    legal = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.afterstates4(0, S(), n, P(cu(b)), P(out), P(reward), P(legal))
    want_legal = np.zeros(n, dtype=np.uint8)
    for a in range(4):
        w = oracle.afterstate_batch(b, np.full(n, a, dtype=np.uint8))
        assert np.array_equal(out[:, a].cpu().numpy(), w['boards']) and np.array_equal(reward[:, a].cpu().numpy(), w['reward']), ('afterstates4', n, a)
        want_legal |= (w['changed'] << a).astype(np.uint8)
    assert np.array_equal(legal.cpu().numpy(), want_legal)
    counts['afterstates4'] += 1


def fuzz_i32():
    n = pick_n()
    b = boards_for(n)
    values = cu(oracle.unpack_values(b).astype(np.int32))
    st = oracle.status_batch(b)
    obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda')
    legal4 = torch.full((n, 4), 9, dtype=torch.uint8, device='cuda')
    done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    lib.status_i32(0, S(), n, P(values), P(obs), P(legal4), P(done))
    assert np.array_equal(obs.cpu().numpy(), st['obs']) and np.array_equal(legal4.cpu().numpy(), ((st['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8))
    assert np.array_equal(done.cpu().numpy(), (st['legal'] == 0).astype(np.uint8))
    acts = rng.integers(0, 4, size=n).astype(np.int32)
    out = torch.full((n, 16), -7, dtype=torch.int32, device='cuda')
    reward = torch.zeros((n,), dtype=torch.int32, device='cuda')
    changed = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    acts_dev = cu(acts)
    lib.afterstate_i32(0, S(), n, P(values), P(acts_dev), 0, P(out), P(reward), P(changed))
    w = oracle.afterstate_batch(b, acts.astype(np.uint8))
    assert np.array_equal(out.cpu().numpy(), oracle.unpack_values(w['boards']).astype(np.int32)) and np.array_equal(reward.cpu().numpy(), w['reward'])
    assert np.array_equal(changed.cpu().numpy(), w['changed'])
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    fresh = torch.zeros((n, 16), dtype=torch.int32, device='cuda')
    lib.reset_keyed_i32_status(0, S(), n, P(cu(keys)), P(fresh), P(obs), P(legal4))
    want = oracle.reset_keyed_i32(keys)
    assert np.array_equal(fresh.cpu().numpy(), np.asarray(want).reshape(n, 16))
    wst = oracle.status_batch(oracle.pack_values(np.asarray(want).reshape(n, 16)).reshape(n, 16))
    assert np.array_equal(obs.cpu().numpy(), wst['obs']) and np.array_equal(legal4.cpu().numpy(), ((wst['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8))
    counts['i32'] += 1


def fuzz_outcomes_and_onehot():
    n = pick_n()
    b = boards_for(n)
    dev = cu(b)
    ob = torch.full((n, 32, 16), 0x33, dtype=torch.uint8, device='cuda')
    pr = torch.full((n, 32), -1.0, dtype=torch.float32, device='cuda')
    ct = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.chance_outcomes(0, S(), n, P(dev), P(ob), P(pr), P(ct))
    w = oracle.chance_outcomes(b)
    assert np.array_equal(ob.cpu().numpy(), w['boards']) and np.array_equal(pr.cpu().numpy(), w['prob']) and np.array_equal(ct.cpu().numpy(), w['count']), ('outcomes', n)
    counts['outcomes'] += 1
    width = int(rng.choice([1, 7, 16, 18, 31, 32, 33, 64]))
    shift = int(rng.choice([0, 0, 4, 8, 16]))  # (4 / 8: not 16-byte aligned -> the dword kernel)
    arena = torch.full((shift + n * 16 * width + 16,), 0x44, dtype=torch.uint8, device='cuda')
    lib.encode_onehot(0, S(), n, P(dev), width, arena[shift:].data_ptr())
    host = arena.cpu().numpy()
    assert np.array_equal(host[shift:shift + n * 16 * width].reshape(n, 16 * width), oracle.encode_onehot(b, width)), ('onehot', n, width, shift)
    assert (host[:shift] == 0x44).all() and (host[shift + n * 16 * width:] == 0x44).all()
    counts['onehot'] += 1


def fuzz_episodes():
    n = int(rng.choice([4, 8, 64, 256, 300, 301, 1024, 1027, int(rng.integers(1, 3000))]))
    stride = n + int(rng.choice([0, 0, 0, 1, 3, 4]))
    ring = int(rng.integers(2, 70))
    rows = int(rng.integers(0, ring + 1))
    first = int(rng.integers(0, ring))
    flags = int(rng.integers(0, 4))
    p_done = float(rng.choice([0.0, 0.01, 0.07, 0.5, 1.0]))
    done = (rng.random((ring, stride)) < p_done).astype(np.uint8) * int(rng.choice([1, 1, 255]))
    reward = rng.integers(0, 70000, size=(ring, stride)).astype(np.int32)
    if flags & 2:
        reward = (reward.astype(np.float32) / 3.0).astype(np.float32)
    max_exp = rng.integers(0, 18, size=(ring, stride)).astype(np.uint8)
    want, want_of = oracle.segment_episodes(done, reward, max_exp, n=n, rows=rows, first_row=first, ring_rows=ring, flags=flags, want_episode_of=True)
    d_done, d_reward, d_max = cu(done), cu(reward), cu(max_exp)
    counts_dev = torch.zeros((n,), dtype=torch.int32, device='cuda')
    lib.count_episodes(0, S(), n, stride, rows, first, ring, P(d_done), flags & 1, P(counts_dev))
    ends = torch.cumsum(counts_dev.to(torch.int64), dim=0)
    offsets = (ends - counts_dev).contiguous()
    assert int(ends[-1]) == len(want), ('count', n, stride, ring, rows, first, flags)
    raw = torch.zeros((max(len(want), 1), 24), dtype=torch.uint8, device='cuda')
    episode_of = torch.full((ring, stride), -7, dtype=torch.int32, device='cuda')
    lib.segment_episodes(0, S(), n, stride, rows, first, ring, P(d_done), P(d_reward), P(d_max), flags, P(offsets), P(raw), P(episode_of))
    got = raw.cpu().numpy()[:len(want)].view(oracle.EPISODE_DTYPE).reshape(-1)
    for field in ('env', 'start', 'length', 'terminated', 'max_exp', 'total_reward'):
        assert np.array_equal(got[field], want[field]), (field, n, stride, ring, rows, first, flags)
    assert np.array_equal(episode_of.cpu().numpy(), want_of), ('episode_of', n, stride, ring, rows, first, flags)
    counts['episodes'] += 1


def fuzz_random_actions():
    n = pick_n()
    base = int(rng.choice([0, 1, 2, 3, 5, int(rng.integers(0, 1 << 40))]))
    shift = int(rng.integers(0, 4))
    seed, step = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))
    arena = torch.full((n + shift + 8,), 0x77, dtype=torch.uint8, device='cuda')
    lib.random_actions(0, S(), n, base, seed, step, arena[shift:].data_ptr())
    host = arena.cpu().numpy()
    m = min(n, 300)  # (the oracle's step word is one ctypes call per environment)
    idx = rng.choice(n, size=m, replace=False) if n > m else np.arange(n)
    assert all(host[shift + i] == (oracle.step_word(seed, base + int(i), step) & 3) for i in idx), ('random_actions', n, base, shift)
    assert (host[:shift] == 0x77).all() and (host[shift + n:] == 0x77).all() and (host[shift:shift + n] < 4).all()
    counts['random_actions'] += 1


t0 = time.time()
jobs = [fuzz_afterstates4, fuzz_i32, fuzz_outcomes_and_onehot, fuzz_episodes, fuzz_random_actions]
while time.time() - t0 < budget:
    jobs[int(rng.integers(0, len(jobs)))]()
torch.cuda.synchronize()
print('all identical:', counts)
