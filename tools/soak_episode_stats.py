#!/usr/bin/env python3
"""One-off soak of env2048_rollout_episodes at scale (GPU box): (a) 8192 environments x 3000 steps in three calls (games span
calls) == the oracle's statistics exactly; (b) 2^20 environments x 1024 steps: the statistics block agrees with the plain
rollout's per-environment counters (finished games, sum of rewards of finished + running games) and boards are identical."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle  # noqa: E402
from simulate_2048_amd import BatchedEnv2048  # noqa: E402

n, seed, base = 8192, 99, 123456
env = BatchedEnv2048(n, seed, device='cuda:0', env_base=base, auto_reset=True)
env.reset()
start = env.boards.cpu().numpy().copy()
stats = None
for T in (700, 1300, 1000):
    stats = env.rollout_episodes(T, stats)
ref = oracle.rollout_episodes(start, seed, 0, 3000, env_base=base, flags=3)
assert np.array_equal(stats.cpu().numpy().view(np.uint64), ref['stats']), 'statistics differ'
assert np.array_equal(env.boards.cpu().numpy(), ref['boards'])
print('slice ok:', BatchedEnv2048.episode_statistics(stats))

N, T = 1 << 20, 1024
a = BatchedEnv2048(N, seed, device='cuda:0', auto_reset=True)
b = BatchedEnv2048(N, seed, device='cuda:0', auto_reset=True)
a.reset()
b.reset()
plain = a.rollout(T)
st = b.rollout_episodes(T)
torch.cuda.synchronize()
w = [int(x) & 0xFFFFFFFFFFFFFFFF for x in st.cpu().tolist()]
assert torch.equal(a.boards, b.boards)
assert w[0] == int(plain['episodes'].sum(dtype=torch.int64)), (w[0], int(plain['episodes'].sum()))
assert w[2] + int(b._ep_return.sum()) == int(plain['return_sum'].sum()), 'reward of finished + running games != total reward'
assert w[1] + int(b._ep_length.sum(dtype=torch.int64)) == N * T
assert sum(w[8:40]) == w[0]
print('full size ok:', BatchedEnv2048.episode_statistics(st))
