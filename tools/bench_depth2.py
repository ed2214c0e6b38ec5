#!/usr/bin/env python3
"""Time env2048_expectimax_depth2 (the whole two-ply expectimax tree per board in one launch) on mid-game boards (GPU box).
Reports microseconds per launch, root boards / s and leaf evaluations / s (every legal (action, cell, tile) successor is
evaluated with two slides: opposite moves score the same, so one horizontal and one vertical slide cover the four moves).  The kernel is pure VALU work (16 B in, 89 B out per board): its roofline is the measured VALU
issue ceiling (profiles/valu_ceiling.json); run it under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES` for the instruction count."""
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402


def main():
    import os

    lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library()  # (an experiment build)
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for lg in (12, 14, 16, 18, 20):
        n = 1 << lg
        b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
        lib.reset(0, s, n, 0, b.data_ptr(), None, 0x2048, 0)
        lib.rollout(0, s, n, 0, b.data_ptr(), 0x2048, 0, 100, None, None, None, 3)
        terms = torch.zeros((n, 4, 4), dtype=torch.int32, device='cuda')
        q = torch.zeros((n, 4), dtype=torch.float32, device='cuda')
        best = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        value = torch.zeros((n,), dtype=torch.float32, device='cuda')
        reps = 20 if lg >= 18 else 100
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                lib.expectimax_depth2(0, s, n, b.data_ptr(), terms.data_ptr(), q.data_ptr(), best.data_ptr(), value.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3 / reps)
        us = statistics.median(times)
        leaves = int(terms[:, :, 3].sum()) * 2  # successors evaluated: 2 tiles per empty cell of every legal afterstate
        out[f'2^{lg}'] = {'us_per_launch': round(us, 2), 'root_boards_per_s': n / us * 1e6, 'successors_per_board': leaves / n,
                          'successor_evaluations_per_s': leaves / us * 1e6, 'leaf_slides_per_s': 2 * leaves / us * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
