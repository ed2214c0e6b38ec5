#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[1] and configs[4] plus host-API overhead.
(bench.py carries the headline configs[2]; this script only prints supporting numbers.)"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from simulate_2048_amd import BatchedEnv2048

SEED = 0x2048
out = {}


def timed(fn, sync=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    if sync:
        torch.cuda.synchronize()
    return time.perf_counter() - t0


# configs[1]: 65 536 boards, cache-resident, launch-latency-bound
env = BatchedEnv2048(1 << 16, SEED, device='cuda:0', auto_reset=True)
env.reset()
gen = torch.Generator().manual_seed(1)
pool = [torch.randint(0, 4, (1 << 16,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]
for t in range(200):
    env.step(pool[t & 7])
K = 3000
dt = timed(lambda: [env.step(pool[t & 7]) for t in range(K)])
out['config1_65536_boards'] = {'us_per_step': dt / K * 1e6, 'board_steps_per_s': (1 << 16) * K / dt,
                               'note': 'cache-resident, launch/host-bound; not an HBM figure'}

# host-API overhead at 2^20: BatchedEnv2048.step() as a user calls it, vs enqueue-only time
n = 1 << 20
env = BatchedEnv2048(n, SEED, device='cuda:0', auto_reset=True)
env.reset()
pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]
for t in range(100):
    env.step(pool[t & 7])
K = 2000
dt = timed(lambda: [env.step(pool[t & 7]) for t in range(K)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(200):
    env.step(pool[t & 7])
enqueue = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
out['api_step_2^20'] = {'us_per_step_end_to_end': dt / K * 1e6, 'host_us_per_call_enqueue_only': enqueue * 1e6,
                        'board_steps_per_s': n * K / dt}

# configs[4]: 2^20 boards x 512 steps with auto-reset: 512 API launches (on-device actions) vs one fused rollout
env_a = BatchedEnv2048(n, SEED, device='cuda:0', auto_reset=True)
env_b = BatchedEnv2048(n, SEED, device='cuda:0', auto_reset=True)
env_a.reset()
env_b.reset()
total = torch.zeros((), dtype=torch.int64, device='cuda')
episodes = torch.zeros((), dtype=torch.int64, device='cuda')


def api_loop():
    global total, episodes
    for _ in range(512):
        _, r, d = env_a.step(None)
        total += r.sum()
        episodes += d.sum()


dt_api = timed(api_loop)
env_c = BatchedEnv2048(n, SEED, device='cuda:0', auto_reset=True)
env_c.reset()
dt_api_pure = timed(lambda: [env_c.step(None) for _ in range(512)])
res = {}
dt_roll = timed(lambda: res.update(env_b.rollout(512)))
same = bool(torch.equal(env_a.boards, env_b.boards)) and int(res['return_sum'].sum()) == int(total) and int(res['episodes'].sum()) == int(episodes)
out['config4_rollout_512'] = {
    'api_512_launches_s': dt_api_pure, 'api_board_steps_per_s': n * 512 / dt_api_pure,
    'api_512_launches_with_torch_reductions_s': dt_api,  # the cross-check loop: + r.sum() and d.sum() per step
    'fused_rollout_s': dt_roll, 'fused_board_steps_per_s': n * 512 / dt_roll,
    'identical_boards_return_episodes': same,
    'episodes': int(episodes), 'mean_episode_length': n * 512 / max(1, int(episodes)),
    'valid_move_fraction': float(res['valid_moves'].sum()) / (n * 512),
    'note': 'fused rollout keeps boards in registers: ALU/issue-bound, no HBM fraction quoted',
}
print(json.dumps(out, indent=1))
