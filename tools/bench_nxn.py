#!/usr/bin/env python3
"""Time env2048_nxn_step (boards of any size S x S, one byte per cell) on mid-game boards (GPU box): K dependent steps of one batch,
random moves drawn in the kernel, auto-reset on.  Reports microseconds per launch, board-steps / s and the rate of the algorithmic
bytes (2 S S board bytes + 4 reward + 1 done per board-step) against the 8 TB/s HBM peak; S = 4 is also run through the packed
env2048_step for comparison."""
import json
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

FLAGS = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS


def timed(fn, reps):
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(reps):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(times)


def main():
    lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library()
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for lg in (16, 20):
        n = 1 << lg
        reward = torch.zeros((n,), dtype=torch.int32, device='cuda')
        done = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        for size in (2, 3, 4, 5, 6, 7, 8):
            b = torch.zeros((n, size * size), dtype=torch.uint8, device='cuda')
            lib.nxn_reset(0, s, n, size, 0, b.data_ptr(), None, 0x2048, 0)
            for k in range(60):  # mid-game boards
                lib.nxn_step(0, s, n, size, 0, b.data_ptr(), None, reward.data_ptr(), done.data_ptr(), None, 0x2048, k, FLAGS)
            step = [100]

            def one(k, b=b, size=size):
                step[0] += 1
                lib.nxn_step(0, s, n, size, 0, b.data_ptr(), None, reward.data_ptr(), done.data_ptr(), None, 0x2048, step[0], FLAGS)

            us = timed(one, 50 if lg >= 20 else 200)
            bytes_per = 2 * size * size + 5
            out[f'2^{lg} S={size}'] = {'us_per_launch': round(us, 2), 'board_steps_per_s': n / us * 1e6, 'bytes_per_board_step': bytes_per,
                                       'algorithmic_TBps': round(n * bytes_per / us * 1e6 / 1e12, 3),
                                       'frac_of_8TBps': round(n * bytes_per / us * 1e6 / 8e12, 3)}
        b4 = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
        lib.reset(0, s, n, 0, b4.data_ptr(), None, 0x2048, 0)
        step = [100]

        def packed(k):
            step[0] += 1
            lib.step(0, s, n, 0, b4.data_ptr(), None, reward.data_ptr(), done.data_ptr(), None, None, 0x2048, step[0], FLAGS)

        us = timed(packed, 200)
        out[f'2^{lg} S=4 packed env2048_step'] = {'us_per_launch': round(us, 2), 'board_steps_per_s': n / us * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
