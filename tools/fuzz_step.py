#!/usr/bin/env python3
"""Randomised differential run of env2048_step / env2048_step_many against the CPU oracle (GPU box): random batch sizes,
global index offsets, seeds, flag combinations (auto-reset, on-device actions, store-policy hints, log-normalised
reward), output subsets (reward, legal, obs, max_exp) and observation formats.  Not part of the test-suite
(the deterministic cases live in tests/test_hip_parity.py); run it after kernel changes:
    python tools/fuzz_step.py [seconds]"""
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
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
t0, cases, boards_total = time.time(), 0, 0
last_report = t0
OBS = [(0, torch.float32), (capi.FLAG_OBS_F16, torch.float16), (capi.FLAG_OBS_BF16, torch.bfloat16)]
while time.time() - t0 < budget:
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, int(rng.integers(1, 40000))]))
    kind = rng.integers(0, 3)
    if kind == 0:
        boards = rng.integers(0, 6, size=(n, 16))
    elif kind == 1:
        boards = np.where(rng.random((n, 16)) < 0.07, 0, rng.integers(1, 4, size=(n, 16)))
    else:
        boards = np.where(rng.random((n, 16)) < 0.3, 0, rng.integers(0, 18, size=(n, 16)))
    boards = boards.astype(np.uint8)
    actions = rng.integers(0, 256, size=n).astype(np.uint8)
    env_base = int(rng.choice([0, 1, 2, 3, int(rng.integers(0, 1 << 40))]))
    seed, step = int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 50))
    flags = int(rng.integers(0, 4))
    want_legal, want_obs, want_reward = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    obs_flag, obs_dtype = OBS[int(rng.integers(0, 3))] if want_obs else OBS[0]
    acts = None if flags & capi.FLAG_RANDOM_ACTIONS else actions
    want_max, norm = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0) and want_reward
    hint = int(rng.choice([0, 0, capi.FLAG_STORES_WRITE_THROUGH, capi.FLAG_STORES_WRITE_BACK]))
    hint |= int(rng.choice([0, capi.FLAG_TILE_SMALL, capi.FLAG_TILE_LARGE]))
    ref = oracle.step_batch_ex(boards, acts, seed, step, env_base=env_base, flags=flags)
    b = torch.as_tensor(boards).cuda()
    a = None if acts is None else torch.as_tensor(acts).cuda()
    reward = torch.full((n,), -7, dtype=torch.float32 if norm else torch.int32, device='cuda') if want_reward else None
    max_exp = torch.full((n,), 99, dtype=torch.uint8, device='cuda') if want_max else None
    done = torch.full((n,), 9, dtype=torch.uint8, device='cuda')
    legal = torch.full((n,), 99, dtype=torch.uint8, device='cuda') if want_legal else None
    obs = torch.full((n, 16), -1.0, dtype=obs_dtype, device='cuda') if want_obs else None
    p = lambda t: None if t is None else t.data_ptr()
    all_flags = flags | obs_flag | hint | (capi.FLAG_REWARD_LOG2_F32 if norm else 0)
    if want_max or rng.integers(0, 2):
        rec = (capi.StepArgs * 1)(capi.StepArgs(n, env_base, p(b), p(a), p(reward), p(done), p(legal), p(obs), seed, step, all_flags, 0, p(max_exp)))
        lib.step_many(0, torch.cuda.current_stream().cuda_stream, rec)
    else:
        lib.step(0, torch.cuda.current_stream().cuda_stream, n, env_base, p(b), p(a), p(reward), p(done), p(legal), p(obs), seed, step,
                 all_flags)
    torch.cuda.synchronize()
    tag = f'n={n} env_base={env_base} seed={seed} step={step} flags={flags} obs={obs_dtype} legal={want_legal} reward={want_reward}'
    np.testing.assert_array_equal(b.cpu().numpy(), ref['boards'], err_msg=tag)
    np.testing.assert_array_equal(done.cpu().numpy(), ref['done'], err_msg=tag)
    tag += f' max_exp={want_max} norm={norm} hint={hint}'
    if want_reward and norm:
        got, want = reward.cpu().numpy(), ref['reward_norm']
        assert np.all(np.abs(got - want) <= np.spacing(np.maximum(want, np.float32(1e-30)))), tag
    elif want_reward:
        np.testing.assert_array_equal(reward.cpu().numpy(), ref['reward'], err_msg=tag)
    if want_max:
        np.testing.assert_array_equal(max_exp.cpu().numpy(), ref['max_exp'], err_msg=tag)
    if want_legal:
        np.testing.assert_array_equal(legal.cpu().numpy(), ref['legal'], err_msg=tag)
    if want_obs:
        np.testing.assert_array_equal(obs.to(torch.float32).cpu().numpy(), ref['obs'], err_msg=tag)
    cases += 1
    boards_total += n
    if time.time() - last_report > 30:  # a long run keeps talking (a silent GPU job is taken for hung)
        last_report = time.time()
        print(f'fuzz_step: {cases} cases so far, all identical', flush=True)
print(f'fuzz_step: {cases} random cases, {boards_total} board-steps, all bit-identical to the oracle')
