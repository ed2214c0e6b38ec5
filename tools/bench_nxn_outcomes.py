#!/usr/bin/env python3
"""Time env2048_nxn_chance_outcomes (every (empty cell, tile) successor of boards of size S x S: 2 S^2 slots of S^2 bytes per board)
on mid-game boards, several batch sizes (GPU box): microseconds per launch and the rate of the bytes written."""
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

lib = capi.library()
s = torch.cuda.current_stream().cuda_stream
out = {}
for size in (4, 6, 8):
    cells = size * size
    for lg in (12, 14, 16, 17):
        n = 1 << lg
        if n * 2 * cells * cells > (3 << 30):
            continue
        b = torch.zeros((n, cells), dtype=torch.uint8, device='cuda')
        lib.nxn_reset(0, s, n, size, 0, b.data_ptr(), None, 1, 0)
        r = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        for k in range(80):
            lib.nxn_step(0, s, n, size, 0, b.data_ptr(), None, r.data_ptr(), d.data_ptr(), None, 1, k + 1, 3)
        slots = torch.empty((n, 2 * cells, cells), dtype=torch.uint8, device='cuda')
        cnt = torch.empty((n,), dtype=torch.uint8, device='cuda')
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                lib.nxn_chance_outcomes(0, s, n, size, b.data_ptr(), slots.data_ptr(), cnt.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 200)
        us = statistics.median(ts)
        out[f'S={size} 2^{lg}'] = {'us_per_launch': round(us, 1), 'bytes_written': n * 2 * cells * cells, 'TB_per_s_written': round(n * 2 * cells * cells / us / 1e6, 3),
                                   'boards_per_s': n / us * 1e6}
        del slots
print(json.dumps(out, indent=1))
