#!/usr/bin/env python3
"""Randomised differential run of the round-3 entry points against the CPU oracle (GPU box):
  * env2048_step_many over RANDOM record sets -- independent batches (shared launches), repeated batches (dependent chains),
    mixed kernel variants, hints, ragged sizes, empty records -- against the oracle applied record by record, in order;
  * env2048_step_state_i32_keyed_status / env2048_reset_keyed_i32_status (fused observation + legal bytes), and the
    reinforce.game mirrors end to end (pooled result sets, env2048_*_mirrored host mirror, np.array reads);
  * env2048_expectimax_depth2 (integer terms, q, best, value);  env2048_random_actions;
  * env2048_step_sequence as two half-batch chains vs ENV2048_FLAG_SINGLE_CHAIN (GPU vs GPU, 2^20-board batches).
python tools/fuzz_round3.py [seconds]"""
import ctypes
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
counts = {'depth_d': 0, 'mirrors': 0, 'step_many': 0, 'records': 0, 'shared_launches': 0, 'keyed_status': 0, 'depth2': 0, 'random_actions': 0, 'two_chains': 0}
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096, 8191]


def boards_for(n):
    kind = rng.integers(0, 3)
    if kind == 0:
        b = rng.integers(0, 6, size=(n, 16))
    elif kind == 1:
        b = np.where(rng.random((n, 16)) < 0.07, 0, rng.integers(1, 4, size=(n, 16)))
    else:
        b = np.where(rng.random((n, 16)) < 0.3, 0, rng.integers(0, 18, size=(n, 16)))
    return b.astype(np.uint8)


def fuzz_step_many():
    n_batches = int(rng.integers(1, 9))
    base_n = int(rng.choice(SIZES + [int(rng.integers(1, 20000))]))
    batches = []
    for _ in range(n_batches):
        n = max(1, int(base_n * rng.uniform(0.55, 1.0))) if rng.random() < 0.8 else int(rng.choice(SIZES))
        batches.append({'n': n, 'host': boards_for(n), 'base': int(rng.choice([0, 1, 3, int(rng.integers(0, 1 << 40))]))})
    for b in batches:
        b['dev'] = torch.from_numpy(b['host']).cuda()
    n_records = int(rng.integers(1, 14))
    variant_pool = [(False, False), (False, False), (True, False), (False, True)]  # (extras, random actions)
    recs, keep, expect = [], [], []
    seed = int(rng.integers(0, 1 << 62))
    for k in range(n_records):
        if rng.random() < 0.1:  # an empty record -- now and then a LONG run of them (round 3's planner hung on 64 in a row)
            for _ in range(int(rng.choice([1, 1, 1, 64, 65, 130]))):
                recs.append(capi.StepArgs(0, 0, None, None, None, None, None, None, seed, k, 0, 0))
            continue
        b = batches[int(rng.integers(0, n_batches))] if rng.random() < 0.35 else batches[k % n_batches]
        extras, random_actions = variant_pool[int(rng.integers(0, len(variant_pool)))] if rng.random() < 0.3 else (False, False)
        n = b['n']
        flags = int(rng.integers(0, 2)) | (capi.FLAG_RANDOM_ACTIONS if random_actions else 0)
        flags |= int(rng.choice([0, 0, 0, capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE]))
        flags |= int(rng.choice([0, 0, 0, capi.FLAG_STORES_WRITE_THROUGH, capi.FLAG_STORES_WRITE_BACK]))
        acts = rng.integers(0, 256, size=n).astype(np.uint8)
        a = torch.from_numpy(acts).cuda()
        reward = torch.full((n,), -5, dtype=torch.int32, device='cuda')
        done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
        legal = torch.full((n,), 99, dtype=torch.uint8, device='cuda') if extras else None
        obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda') if extras and rng.random() < 0.5 else None
        step = int(rng.integers(0, 1 << 40))
        recs.append(capi.StepArgs(n, b['base'], P(b['dev']), None if random_actions else P(a), P(reward), P(done), P(legal), P(obs), seed, step, flags, 0))
        keep.append((a, reward, done, legal, obs))
        # the oracle, record by record in order
        ref = oracle.step_batch_ex(b['host'], None if random_actions else acts, seed, step, env_base=b['base'], flags=flags & 3)
        b['host'] = ref['boards']
        expect.append((reward, done, legal, obs, ref))
    arr = (capi.StepArgs * len(recs))(*recs)
    plan = lib.step_many_plan(arr)
    lib.step_many(0, S(), arr)
    torch.cuda.synchronize()
    for b in batches:
        np.testing.assert_array_equal(b['dev'].cpu().numpy(), b['host'], err_msg=f'step_many boards, plan {plan["launch_of"]}')
    # outputs of a record survive only if no later record of the call wrote the same arrays: every record has its own here
    for reward, done, legal, obs, ref in expect:
        np.testing.assert_array_equal(reward.cpu().numpy(), ref['reward'])
        np.testing.assert_array_equal(done.cpu().numpy(), ref['done'])
        if legal is not None:
            np.testing.assert_array_equal(legal.cpu().numpy(), ref['legal'])
        if obs is not None:
            np.testing.assert_array_equal(obs.cpu().numpy(), ref['obs'])
    live = [r for r in recs if r.n]
    counts['step_many'] += 1
    counts['records'] += len(live)
    counts['shared_launches'] += len(live) - plan['launches']


def fuzz_keyed_status():
    n = int(rng.choice(SIZES + [int(rng.integers(1, 9000))]))
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    values = oracle.unpack_values(boards_for(n)).astype(np.int32)
    actions = rng.integers(-3, 9, size=n).astype(np.int32)
    sc = rng.integers(0, 500, size=n).astype(np.int32)
    done = (rng.random(n) < 0.2).astype(np.uint8)
    total = rng.integers(0, 5000, size=n).astype(np.float32)
    ref = oracle.step_state_i32_keyed(values, actions, sc, done, total, keys)
    st = oracle.status_batch(oracle.pack_values(ref['boards']).reshape(n, 16))
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    k, bi, a, si, di, ti = cu(keys.view(np.int32)), cu(values), cu(actions), cu(sc), cu(done), cu(total)
    bo = torch.zeros((n, 16), dtype=torch.int32, device='cuda')
    so = torch.zeros((n,), dtype=torch.int32, device='cuda')
    do_ = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    to_ = torch.zeros((n,), dtype=torch.float32, device='cuda')
    rw = torch.zeros((n,), dtype=torch.float32, device='cuda')
    mt = torch.zeros((n,), dtype=torch.int32, device='cuda')
    obs = torch.full((n, 16), -1.0, dtype=torch.float32, device='cuda')
    l4 = torch.full((n, 4), 9, dtype=torch.uint8, device='cuda')
    lib.step_state_i32_keyed_status(0, S(), n, P(k), P(bi), P(bo), P(a), P(si), P(so), P(di), P(do_), P(ti), P(to_), P(rw), P(mt), P(obs), P(l4))
    torch.cuda.synchronize()
    for got, key in ((bo, 'boards'), (so, 'step_count'), (do_, 'done'), (to_, 'total_reward'), (rw, 'reward'), (mt, 'max_tile')):
        np.testing.assert_array_equal(got.cpu().numpy(), ref[key], err_msg=f'keyed_status {key} n={n}')
    np.testing.assert_array_equal(obs.cpu().numpy(), st['obs'])
    np.testing.assert_array_equal(l4.cpu().numpy(), ((st['legal'][:, None] >> np.arange(4)) & 1).astype(np.uint8))
    fresh = torch.zeros((n, 16), dtype=torch.int32, device='cuda')
    lib.reset_keyed_i32_status(0, S(), n, P(k), P(fresh), P(obs), P(l4))
    torch.cuda.synchronize()
    want = oracle.reset_keyed_i32(keys)
    np.testing.assert_array_equal(fresh.cpu().numpy(), want)
    np.testing.assert_array_equal(obs.cpu().numpy(), oracle.status_batch(oracle.pack_values(want).reshape(n, 16))['obs'])
    counts['keyed_status'] += 1


def fuzz_mirrors():
    """The reinforce.game mirrors end to end at small batch sizes: result sets from the pool, kernels that also write the
    pinned host mirror, np.array(...) reads through it -- against the oracle and against plain device-to-host copies."""
    from simulate_2048_amd.reinforce.game import env as genv

    n = int(rng.integers(1, 460))  # around the mirror size limit (N ~ 430) as well
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    states = genv.batched_reset(keys)
    want = oracle.reset_keyed_i32(keys)
    np.testing.assert_array_equal(np.array(states.board).reshape(n, 16), want)
    assert not np.array(states.step_count).any() and not np.array(states.done).any() and not np.array(states.total_reward).any()
    for move in range(int(rng.integers(1, 6))):
        before = (np.array(states.board).reshape(n, 16), np.array(states.step_count), np.array(states.done).astype(np.uint8),
                  np.array(states.total_reward))
        actions = rng.integers(0, 4, size=n).astype(np.int32)
        keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
        states, rewards, dones, info = genv.batched_step(states, actions, keys)
        ref = oracle.step_state_i32_keyed(before[0], actions, before[1], before[2], before[3], keys)
        st = oracle.status_batch(oracle.pack_values(ref['boards']).reshape(n, 16))
        fields = ((states.board, ref['boards'].reshape(n, 4, 4)), (states.step_count, ref['step_count']), (dones, ref['done'].astype(bool)),
                  (states.total_reward, ref['total_reward']), (rewards, ref['reward']), (info['max_tile'], ref['max_tile']),
                  (genv.batched_get_observation(states), st['obs']),
                  (genv.batched_get_legal_actions(states), ((st['legal'][:, None] >> np.arange(4)) & 1).astype(bool)))
        for t, expect in fields:
            np.testing.assert_array_equal(np.array(t), expect, err_msg=f'mirrors n={n} move={move}')
            np.testing.assert_array_equal(t.as_subclass(torch.Tensor).cpu().numpy(), expect)
    counts['mirrors'] += 1


def fuzz_depth2():
    n = int(rng.choice(SIZES[:9] + [int(rng.integers(1, 1500))]))
    boards = boards_for(n)
    b = torch.from_numpy(boards).cuda()
    terms = torch.zeros((n, 4, 4), dtype=torch.int32, device='cuda')
    q = torch.zeros((n, 4), dtype=torch.float32, device='cuda')
    best = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    value = torch.zeros((n,), dtype=torch.float32, device='cuda')
    lib.expectimax_depth2(0, S(), n, P(b), P(terms), P(q), P(best), P(value))
    torch.cuda.synchronize()
    ref = oracle.expectimax_depth2(boards)
    np.testing.assert_array_equal(terms.cpu().numpy(), ref['terms'])
    np.testing.assert_array_equal(q.cpu().numpy(), ref['q'].astype(np.float32))
    np.testing.assert_array_equal(best.cpu().numpy(), ref['best'])
    np.testing.assert_array_equal(value.cpu().numpy(), ref['value'].astype(np.float32))
    counts['depth2'] += 1


def fuzz_depth_d():
    """ops.expectimax at depth 3 (sometimes 4) on a few random boards: dense frontiers + depth-2 kernel + back-ups == the oracle's
    recursion, bit for bit."""
    from simulate_2048_amd import ops, runtime

    depth = 4 if rng.random() < 0.15 else 3
    n = int(rng.integers(1, 5 if depth == 4 else 40))
    boards = boards_for(n)
    out = ops.expectimax(torch.from_numpy(boards).cuda(), depth, rt=runtime.Runtime(lib, 'cuda:0'))
    ref = oracle.expectimax_depth(boards, depth)
    np.testing.assert_array_equal(out['q'].cpu().numpy(), ref['q'])
    np.testing.assert_array_equal(out['best'].cpu().numpy(), ref['best'])
    np.testing.assert_array_equal(out['value'].cpu().numpy(), ref['value'])
    counts['depth_d'] += 1


def fuzz_random_actions():
    n = int(rng.choice(SIZES + [int(rng.integers(1, 5000))]))
    base, seed, step = int(rng.choice([0, 1, 3, int(rng.integers(0, 1 << 40))])), int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))
    out = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    lib.random_actions(0, S(), n, base, seed, step, P(out))
    boards = boards_for(n)
    b1, b2 = torch.from_numpy(boards).cuda(), torch.from_numpy(boards).cuda()
    r1, r2 = (torch.zeros((n,), dtype=torch.int32, device='cuda') for _ in range(2))
    d1, d2 = (torch.zeros((n,), dtype=torch.uint8, device='cuda') for _ in range(2))
    lib.step(0, S(), n, base, P(b1), P(out), P(r1), P(d1), None, None, seed, step, 1)
    lib.step(0, S(), n, base, P(b2), None, P(r2), P(d2), None, None, seed, step, 3)
    torch.cuda.synchronize()
    assert torch.equal(b1, b2) and torch.equal(r1, r2) and torch.equal(d1, d2) and int(out.max()) <= 3
    pick = rng.integers(0, n, size=min(n, 20))
    assert all(int(out[i]) == (oracle.step_word(seed, base + int(i), step) & 3) for i in pick)
    counts['random_actions'] += 1


def fuzz_two_chains():
    n = (1 << 20) + int(rng.choice([0, 0, 77, 2048, 2049, 123457]))
    K = int(rng.integers(32, 40))
    base, seed = int(rng.choice([0, 5, int(rng.integers(0, 1 << 40))])), int(rng.integers(0, 1 << 62))
    start = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, S(), n, base, P(start), None, seed, 0)
    lib.rollout(0, S(), n, base, P(start), seed, 0, int(rng.integers(20, 150)), None, None, None, 3)
    random_actions = bool(rng.integers(0, 2))
    acts = torch.randint(0, 4, (8, n), dtype=torch.uint8, device='cuda')
    table = None if random_actions else (ctypes.c_void_p * K)(*[acts[k & 7].data_ptr() for k in range(K)])
    flags = int(rng.integers(0, 2)) | (capi.FLAG_RANDOM_ACTIONS if random_actions else 0)
    stride = int(rng.choice([0, n]))
    out = []
    for extra in (0, capi.FLAG_SINGLE_CHAIN):
        b = start.clone()
        r = torch.zeros((K if stride else 1, n), dtype=torch.int32, device='cuda')
        d = torch.zeros((K if stride else 1, n), dtype=torch.uint8, device='cuda')
        lib.step_sequence(0, S(), n, base, P(b), table, P(r), P(d), None, None, stride, seed, 1000, K, flags | extra)
        out.append((b, r, d))
    torch.cuda.synchronize()
    assert lib.step_sequence_plan(n, K, flags)['chains'] == 2
    for x, y in zip(*out):
        assert torch.equal(x, y), 'two chains != single chain'
    counts['two_chains'] += 1


CASES = [(fuzz_step_many, 10), (fuzz_mirrors, 3), (fuzz_keyed_status, 3), (fuzz_depth2, 3), (fuzz_depth_d, 2), (fuzz_random_actions, 2), (fuzz_two_chains, 1)]
weights = np.array([w for _, w in CASES], dtype=float)
weights /= weights.sum()
while time.time() - t0 < budget:
    CASES[int(rng.choice(len(CASES), p=weights))][0]()
    if time.time() - last > 30:
        last = time.time()
        print('fuzz_round3:', counts, flush=True)
print('fuzz_round3 done, all identical:', counts)
