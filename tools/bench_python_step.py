#!/usr/bin/env python3
"""What a PYTHON loop over BatchedEnv2048.step / TrajectoryBuffer.step costs per step next to the C-level sequence call
(env2048_step_sequence, what bench.py times): microseconds per step, wall clock, K = 200 steps between two synchronisations."""
import json
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import BatchedEnv2048, TrajectoryBuffer  # noqa: E402

out = {}
K = 200
for lg in (12, 16, 20):
    n = 1 << lg
    env = BatchedEnv2048(n, seed=1, device='cuda:0', auto_reset=True)
    env.reset()
    env.rollout(64)
    acts = [torch.randint(0, 4, (n,), dtype=torch.uint8, device='cuda') for _ in range(8)]

    def timed(fn):
        fn()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        return round(statistics.median(ts), 2)

    def python_loop():
        for k in range(K):
            env.step(acts[k & 7])

    def python_loop_random():
        for k in range(K):
            env.step(None)

    row = {'BatchedEnv2048.step(actions) in a Python loop': timed(python_loop), 'BatchedEnv2048.step(None) in a Python loop': timed(python_loop_random),
           'BatchedEnv2048.step_many (one C call for the K steps)': timed(lambda: env.step_many(torch.stack([acts[k & 7] for k in range(K)])))}
    full = BatchedEnv2048(n, seed=1, device='cuda:0', auto_reset=True, with_legal=True, with_obs=True)
    full.reset()
    traj = TrajectoryBuffer(full, horizon=K, ring=True)
    traj.begin()

    def traj_loop():
        for k in range(K):
            traj.step(None)

    row['TrajectoryBuffer.step(None) (+ legal + obs rows) in a Python loop'] = timed(traj_loop)
    out[f'2^{lg}'] = row
    print(f'2^{lg}', row, flush=True)
print(json.dumps(out, indent=1))
