#!/usr/bin/env python3
"""Time the trajectory post-processing kernels piece by piece (GPU box): env2048_count_episodes, env2048_segment_episodes and the
whole TrajectoryBuffer.episodes() on a window of T recorded steps of N games (default 256 x 2^20): microseconds and the rate of the
row bytes each kernel reads (done 1 B; + reward 4 B + max_exp 1 B for the segmentation)."""
import json
import os
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import BatchedEnv2048, TrajectoryBuffer, capi  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
env = BatchedEnv2048(N, seed=3, device='cuda:0', auto_reset=True)
env.reset()
traj = TrajectoryBuffer(env, horizon=T, record_obs=False, record_legal=False)
traj.begin()
for t in range(T):
    traj.step(None)
torch.cuda.synchronize()
lib = env.lib
S = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


p = lambda t: None if t is None else t.data_ptr()
counts = torch.zeros((N,), dtype=torch.int32, device='cuda')
first, count = traj.window()
first_row = traj._row(first)
out = {'rows': T, 'envs': N}
us = timed(lambda: lib.count_episodes(0, S, N, N, count, first_row, T, p(traj.done), 0, p(counts)))
out['count_episodes'] = {'us': round(us, 1), 'TB_per_s_rows_read': round(T * N * 1 / us / 1e6, 3)}
ends = torch.cumsum(counts.to(torch.int64), dim=0)
offsets = (ends - counts).contiguous()
total = int(ends[-1])
raw = torch.zeros((total, 24), dtype=torch.uint8, device='cuda')
us = timed(lambda: lib.segment_episodes(0, S, N, N, count, first_row, T, p(traj.done), p(traj.reward), p(traj.max_exp), 0, p(offsets), p(raw), None))
out['segment_episodes'] = {'us': round(us, 1), 'episodes': total, 'TB_per_s_rows_read': round(T * N * 6 / us / 1e6, 3)}
episode_of = torch.empty((T, N), dtype=torch.int32, device='cuda')
us = timed(lambda: lib.segment_episodes(0, S, N, N, count, first_row, T, p(traj.done), p(traj.reward), p(traj.max_exp), 0, p(offsets), p(raw), p(episode_of)))
out['segment_episodes + episode_of'] = {'us': round(us, 1), 'TB_per_s_rows_read_and_written': round(T * N * 10 / us / 1e6, 3)}
for _ in range(3):  # (the first calls grow the caching allocator's pool)
    traj.episodes()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    traj.episodes()
torch.cuda.synchronize()
out['TrajectoryBuffer.episodes() wall ms'] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
print(json.dumps(out, indent=1))
