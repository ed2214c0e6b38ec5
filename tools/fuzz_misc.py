#!/usr/bin/env python3
"""Randomised differential run of the round-2 kernels against the CPU oracle (GPU box): env2048_replay vs single steps,
env2048_count/segment_episodes on random done/reward/max_exp rows (ring offsets, frozen and auto-reset forms),
env2048_expectimax_ply, env2048_step_state_i32_keyed / env2048_reset_keyed_i32, env2048_rollout_episodes -- random sizes
(ragged around the 64- / 256- / 2048-board tiles), both tilings.    python tools/fuzz_misc.py [seconds]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import oracle
from simulate_2048_amd import capi

lib = capi.library()
rng = np.random.default_rng(int(time.time()))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = last = time.time()
counts = {'replay': 0, 'episodes': 0, 'ply': 0, 'keyed': 0, 'rollout_stats': 0}
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049]


def size():
    return int(rng.choice(SIZES + [int(rng.integers(1, 9000))]))


def boards_for(n):
    kind = rng.integers(0, 3)
    if kind == 0:
        b = rng.integers(0, 6, size=(n, 16))
    elif kind == 1:
        b = np.where(rng.random((n, 16)) < 0.07, 0, rng.integers(1, 4, size=(n, 16)))
    else:
        b = np.where(rng.random((n, 16)) < 0.3, 0, rng.integers(0, 18, size=(n, 16)))
    return b.astype(np.uint8)


while time.time() - t0 < budget:
    which = rng.integers(0, 5)
    n = size()
    tile = int(rng.choice([0, capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE]))
    seed, base = int(rng.integers(0, 1 << 62)), int(rng.choice([0, 1, 3, int(rng.integers(0, 1 << 40))]))
    if which == 0:  # replay == single steps (oracle)
        T = int(rng.integers(1, 12))
        stride = n + int(rng.choice([0, 0, 3, 4, 64]))
        auto = int(rng.integers(0, 2))
        start = boards_for(n)
        acts = rng.integers(0, 256, size=(T, stride)).astype(np.uint8)
        b = torch.from_numpy(start).cuda()
        a = torch.from_numpy(acts).cuda()
        rew = torch.full((T, stride), -9, dtype=torch.int32, device='cuda')
        don = torch.full((T, stride), 9, dtype=torch.uint8, device='cuda')
        ret = torch.zeros((n,), dtype=torch.int64, device='cuda')
        eps = torch.zeros((n,), dtype=torch.int32, device='cuda')
        lib.replay(0, S(), n, base, P(b), P(a), P(rew), P(don), stride, seed, 7, T, auto | tile, P(ret), P(eps), None)
        ref = start
        tot = np.zeros(n, np.int64)
        for t in range(T):
            r = oracle.step_batch(ref, acts[t, :n].copy(), seed, 7 + t, env_base=base, flags=auto)
            ref = r['boards']
            tot += r['reward']
            assert np.array_equal(rew[t, :n].cpu().numpy(), r['reward']) and np.array_equal(don[t, :n].cpu().numpy(), r['done']), ('replay', n, T, stride, tile)
        assert np.array_equal(b.cpu().numpy(), ref) and np.array_equal(ret.cpu().numpy(), tot)
        assert bool((rew[:, n:] == -9).all()) and bool((don[:, n:] == 9).all())
        counts['replay'] += 1
    elif which == 1:  # episode segmentation
        C = int(rng.integers(2, 40))
        rows = int(rng.integers(0, C + 1))
        first = int(rng.integers(0, C))
        stride = n + int(rng.choice([0, 5]))
        flags = int(rng.integers(0, 2))
        done = (rng.random((C, stride)) < rng.choice([0.02, 0.2, 0.6])).astype(np.uint8)
        f32 = bool(rng.integers(0, 2))
        reward = (rng.random((C, stride)).astype(np.float32) if f32 else rng.integers(0, 5000, size=(C, stride)).astype(np.int32))
        mx = rng.integers(0, 18, size=(C, stride)).astype(np.uint8)
        fl = flags | (2 if f32 else 0)
        want, want_of = oracle.segment_episodes(done, reward, mx, n=n, rows=rows, first_row=first, ring_rows=C, flags=fl, want_episode_of=True)
        d, r, m = torch.from_numpy(done).cuda(), torch.from_numpy(reward).cuda(), torch.from_numpy(mx).cuda()
        cnt = torch.zeros((n,), dtype=torch.int32, device='cuda')
        lib.count_episodes(0, S(), n, stride, rows, first, C, P(d), fl, P(cnt))
        ends = torch.cumsum(cnt.to(torch.int64), 0)
        off = (ends - cnt).contiguous()
        total = int(ends[-1])
        assert total == len(want), ('count', n, C, rows, first, fl)
        raw = torch.zeros((max(total, 1), 24), dtype=torch.uint8, device='cuda')
        eo = torch.full((C, stride), -7, dtype=torch.int32, device='cuda')
        lib.segment_episodes(0, S(), n, stride, rows, first, C, P(d), P(r), P(m), fl, P(off), P(raw), P(eo))
        got = raw[:total].cpu().numpy().view(oracle.EPISODE_DTYPE).reshape(-1)
        assert np.array_equal(got, want), ('segment', n, C, rows, first, fl)
        assert np.array_equal(eo.cpu().numpy(), want_of), ('episode_of', n, C, rows, first, fl)
        counts['episodes'] += 1
    elif which == 2:  # expectimax ply
        start = boards_for(n)
        b = torch.from_numpy(start).cuda()
        sb = torch.full((n, 4, 32, 16), 99, dtype=torch.uint8, device='cuda')
        sp = torch.full((n, 4, 32), -1.0, dtype=torch.float32, device='cuda')
        ss = torch.full((n, 4, 32), 99, dtype=torch.uint8, device='cuda')
        sc = torch.full((n, 4), 99, dtype=torch.uint8, device='cuda')
        ar = torch.full((n, 4), -1, dtype=torch.int32, device='cuda')
        lg = torch.full((n,), 99, dtype=torch.uint8, device='cuda')
        lib.expectimax_ply(0, S(), n, P(b), P(sb), P(sp), P(ss), P(sc), P(ar), P(lg))
        ref = oracle.expectimax_ply(start)
        for got, key in ((sb, 'boards'), (sp, 'prob'), (ss, 'status'), (sc, 'count'), (ar, 'reward'), (lg, 'legal')):
            assert np.array_equal(got.cpu().numpy(), ref[key]), ('ply', key, n)
        counts['ply'] += 1
    elif which == 3:  # keyed step / reset on int32 boards
        keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
        k = torch.from_numpy(keys.view(np.int32)).cuda()
        fresh = torch.zeros((n, 16), dtype=torch.int32, device='cuda')
        lib.reset_keyed_i32(0, S(), n, P(k), P(fresh))
        assert np.array_equal(fresh.cpu().numpy(), oracle.reset_keyed_i32(keys))
        values = oracle.unpack_values(boards_for(n)).astype(np.int32)
        actions = rng.integers(-5, 9, size=n).astype(np.int32)
        sc0 = rng.integers(0, 99, size=n).astype(np.int32)
        d0 = (rng.random(n) < 0.2).astype(np.uint8)
        t0_ = rng.integers(0, 999, size=n).astype(np.float32)
        ref = oracle.step_state_i32_keyed(values, actions, sc0, d0, t0_, keys)
        bi, a, si, di, ti = (torch.from_numpy(x).cuda() for x in (values, actions, sc0, d0, t0_))
        bo, so, do_, to_ = torch.zeros_like(bi), torch.zeros_like(si), torch.zeros_like(di), torch.zeros_like(ti)
        rw = torch.zeros((n,), dtype=torch.float32, device='cuda')
        mt = torch.zeros((n,), dtype=torch.int32, device='cuda')
        lib.step_state_i32_keyed(0, S(), n, P(k), P(bi), P(bo), P(a), P(si), P(so), P(di), P(do_), P(ti), P(to_), P(rw), P(mt))
        for got, key in ((bo, 'boards'), (so, 'step_count'), (do_, 'done'), (to_, 'total_reward'), (rw, 'reward'), (mt, 'max_tile')):
            assert np.array_equal(got.cpu().numpy(), ref[key]), ('keyed', key, n)
        counts['keyed'] += 1
    else:  # rollout with episode statistics, both reset modes, games spanning two calls
        T1, T2 = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        auto = int(rng.integers(0, 2))
        start = boards_for(n)
        b = torch.from_numpy(start).cuda()
        el = torch.zeros((n,), dtype=torch.int32, device='cuda')
        er = torch.zeros((n,), dtype=torch.int64, device='cuda')
        st = torch.zeros((40,), dtype=torch.int64, device='cuda')
        st[4] = -1
        lib.rollout_episodes(0, S(), n, base, P(b), seed, 3, T1, 2 | auto | tile, P(el), P(er), P(st))
        lib.rollout_episodes(0, S(), n, base, P(b), seed, 3 + T1, T2, 2 | auto | tile, P(el), P(er), P(st))
        ref = oracle.rollout_episodes(start, seed, 3, T1 + T2, env_base=base, flags=2 | auto)
        assert np.array_equal(st.cpu().numpy().view(np.uint64), ref['stats']), ('rollout_stats', n, T1, T2, auto, tile)
        assert np.array_equal(b.cpu().numpy(), ref['boards']) and np.array_equal(el.cpu().numpy().view(np.uint32), ref['ep_length'])
        assert np.array_equal(er.cpu().numpy(), ref['ep_return'])
        counts['rollout_stats'] += 1
    if time.time() - last > 30:
        last = time.time()
        print(f'fuzz_misc: {counts} so far, all identical', flush=True)
print(f'fuzz_misc: {counts} random cases, all identical to the oracle')
