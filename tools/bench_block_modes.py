#!/usr/bin/env python3
"""A K-step timed block (barrier-free, one GPU) three ways: K launches enqueued by env2048_step_sequence, the same K launches
replayed as one hipGraph (device step counter), wall clock per step between two synchronisations -- the shape of bench.py's
timed region, to see what a 20-step block costs beyond its kernels."""
import ctypes
import json
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import BatchedEnv2048, capi  # noqa: E402


def main():
    n = 1 << 20
    out = {}
    lib = capi.library()
    for K in (20, 200):
        env = BatchedEnv2048(n, 0x2048, device='cuda:0', auto_reset=True, device_step_counter=True)
        env.reset()
        env.rollout(64)
        env.step_index = 1000
        gen = torch.Generator(device='cpu').manual_seed(1)
        pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for t in range(K):
                    env.step(pool[t & 7])
                env.commit_steps()
        torch.cuda.synchronize()
        table = (ctypes.c_void_p * K)(*[pool[t & 7].data_ptr() for t in range(K)])
        h = torch.cuda.current_stream().cuda_stream
        eager_env = BatchedEnv2048(n, 0x2048, device='cuda:0', auto_reset=True)
        eager_env.boards.copy_(env.boards)
        step = [5000]

        def eager():
            lib.step_sequence(0, h, n, 0, eager_env.boards.data_ptr(), table, eager_env.reward.data_ptr(), eager_env.done.data_ptr(), None,
                              None, 0, 0x2048, step[0], K, 1)
            step[0] += K

        res = {}
        for name, fn in (('eager_step_sequence', eager), ('graph_replay', graph.replay)):
            for _ in range(5):
                fn()
            walls = []
            for _ in range(200 if K == 20 else 40):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) / K * 1e6)
            res[name] = {'wall_us_per_step_median': round(statistics.median(walls), 3), 'min': round(min(walls), 3)}
        out[f'K={K}'] = res
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
