#!/usr/bin/env python3
"""Time env2048_afterstates4 (the four afterstates + merge scores + legal mask of every board, one launch) on mid-game boards
(GPU box): microseconds per launch and the rate of its algorithmic bytes (16 in, 64 + 16 + 1 out per board)."""
import json
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library()
s = torch.cuda.current_stream().cuda_stream
out = {}
for lg in (14, 16, 18, 20, 22):
    n = 1 << lg
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, s, n, 0, b.data_ptr(), None, 0x2048, 0)
    lib.rollout(0, s, n, 0, b.data_ptr(), 0x2048, 0, 100, None, None, None, 3)
    ob = torch.empty((n, 4, 16), dtype=torch.uint8, device='cuda')
    rw = torch.empty((n, 4), dtype=torch.int32, device='cuda')
    lg_ = torch.empty((n,), dtype=torch.uint8, device='cuda')
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            lib.afterstates4(0, s, n, b.data_ptr(), ob.data_ptr(), rw.data_ptr(), lg_.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    us = statistics.median(ts)
    out[f'2^{lg}'] = {'us_per_launch': round(us, 2), 'boards_per_s': n / us * 1e6, 'TB_per_s': round(n * 97 / us / 1e6, 3)}
print(json.dumps(out, indent=1))
