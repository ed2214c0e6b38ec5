#!/usr/bin/env python3
"""Measurements for the SURVEY 8(f) 'next' rows (N1-N4) on one GPU: supporting numbers, not the headline."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from simulate_2048_amd import BatchedEnv2048, ops
from simulate_2048_amd.reinforce.game import env as genv
from simulate_2048_amd.trajectory import TrajectoryBuffer

out = {}


def per_call(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


n = 1 << 20
# N1: fused legal + obs
for name, kw in (('plain', {}), ('legal', {'with_legal': True}), ('legal_obs', {'with_legal': True, 'with_obs': True})):
    e = BatchedEnv2048(n, 1, device='cuda:0', auto_reset=True, **kw)
    e.reset()
    for _ in range(64):
        e.step(None)
    acts = torch.randint(0, 4, (n,), dtype=torch.uint8).cuda()
    dt = per_call(lambda: e.step(acts), 1000)
    nbytes = 38 + (1 if 'legal' in name else 0) + (64 if 'obs' in name else 0)
    out[f'N1_step_{name}'] = {'us_per_step': dt * 1e6, 'GB_per_s': nbytes * n / dt / 1e9, 'bytes_per_board': nbytes}

# N1 with 16-bit observations (f16 / bf16): the same kernel, half the observation bytes
for name, dt_ in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    e16 = BatchedEnv2048(n, 7, device='cuda:0', auto_reset=True, with_legal=True, with_obs=True, obs_dtype=dt_)
    e16.reset()
    for _ in range(64):
        e16.step(None)
    acts16 = torch.randint(0, 4, (n,), dtype=torch.uint8).cuda()
    dt = per_call(lambda: e16.step(acts16), 300)
    out[f'N1_step_legal_obs_{name}'] = {'us_per_step': dt * 1e6, 'GB_per_s': (39 + 32) * n / dt / 1e9, 'bytes_per_board': 71}

# N2: reinforce.game batched_step (one fused kernel) at the reference's batch size and at 2^20
for nn in (8, 1 << 20):
    keys = np.stack([np.zeros(nn, dtype=np.uint32), np.arange(nn, dtype=np.uint32)], axis=1)
    st = genv.batched_reset(keys)
    acts = torch.randint(0, 4, (nn,), dtype=torch.int32).cuda()
    dt = per_call(lambda: genv.batched_step(st, acts, keys), 200)
    out[f'N2_batched_step_numpy_keys_n{nn}'] = {'us_per_call': dt * 1e6, 'board_steps_per_s': nn / dt}
    keys_dev = torch.from_numpy(keys.view(np.int32)).cuda()  # keys already on the device (what a DLPack hand-over gives)
    dt = per_call(lambda: genv.batched_step(st, acts, keys_dev), 200)
    out[f'N2_batched_step_device_keys_n{nn}'] = {'us_per_call': dt * 1e6, 'board_steps_per_s': nn / dt}
    dt = per_call(lambda: genv.batched_get_observation(st), 200)
    out[f'N2_batched_get_observation_n{nn}'] = {'us_per_call': dt * 1e6}
    dt = per_call(lambda: genv.batched_get_legal_actions(st), 200)
    out[f'N2_batched_get_legal_actions_n{nn}'] = {'us_per_call': dt * 1e6}

# N3: all chance outcomes of 2^18 afterstates (512 B out per board)
e = BatchedEnv2048(1 << 18, 2, device='cuda:0', auto_reset=True)
e.reset()
e.rollout(80)
dt = per_call(lambda: ops.chance_outcomes(e.boards), 50)
out['N3_chance_outcomes_2^18'] = {'us_per_call': dt * 1e6, 'boards_per_s': (1 << 18) / dt, 'GB_per_s': (16 + 512 + 128 + 1) * (1 << 18) / dt / 1e9}

dt = per_call(lambda: ops.afterstates4(e.boards), 50)
out['N3_afterstates4_2^18'] = {'us_per_call': dt * 1e6, 'boards_per_s': (1 << 18) / dt, 'GB_per_s': (16 + 64 + 16 + 1) * (1 << 18) / dt / 1e9}

# N4: trajectory recording (legal+obs rows written by the kernel) and the one-hot transform
e = BatchedEnv2048(n, 3, device='cuda:0', auto_reset=True)
e.reset()
traj = TrajectoryBuffer(e, 64)
traj.begin()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(64):
    traj.step(None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 64
out['N4_trajectory_step_2^20'] = {'us_per_step': dt * 1e6, 'board_steps_per_s': n / dt}
dt = per_call(lambda: ops.onehot(e.boards[: 1 << 18], 31), 50)
out['N4_onehot31_2^18'] = {'us_per_call': dt * 1e6, 'GB_per_s': (16 + 496) * (1 << 18) / dt / 1e9}

# N4 (round 2): recording with max-tile rows, episode segmentation of a [512, 2^20]-row recording, rollout statistics
e = BatchedEnv2048(n, 4, device='cuda:0', auto_reset=True)
e.reset()
T = 256
traj = TrajectoryBuffer(e, T, record_obs=False, record_legal=False)
traj.begin()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(T):
    traj.step(None)
torch.cuda.synchronize()
out['N4_record_reward_done_maxexp_2^20'] = {'us_per_step': (time.perf_counter() - t0) / T * 1e6}
traj.episodes()  # (the first call pays torch's lazy loading of cumsum / where / shift kernels: ~0.1-0.4 s once per process)
torch.cuda.synchronize()
t0 = time.perf_counter()
ep = traj.episodes()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out['N4_episodes_256x2^20_rows'] = {'ms': dt * 1e3, 'records': int(ep['env'].shape[0]), 'rows_per_s': T * n / dt,
                                   'GB_per_s_rows_read_twice': T * n * (1 + 1 + 4 + 1) / dt / 1e9}
e = BatchedEnv2048(n, 5, device='cuda:0', auto_reset=True)
e.reset()
dt = per_call(lambda: e.rollout(512), 3)
dt_stats = per_call(lambda: e.rollout_episodes(512), 3)
out['N4_rollout_512_plain_vs_episode_stats'] = {'ms_plain': dt * 1e3, 'ms_with_episode_statistics': dt_stats * 1e3}
print(json.dumps(out, indent=1))
