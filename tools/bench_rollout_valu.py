#!/usr/bin/env python3
"""The fused rollout (boards in registers, no memory traffic in the loop) as a measure of what VALU issue rate the game logic's
REAL instruction stream reaches: time per 2^20-board step in both tilings and at several occupancies.  Run under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY` for the instruction counts."""
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

lib = capi.library()
s = torch.cuda.current_stream().cuda_stream
T = 64
out = {}
for lg in (17, 18, 19, 20, 21):
    n = 1 << lg
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, s, n, 0, b.data_ptr(), None, 7, 0)
    lib.rollout(0, s, n, 0, b.data_ptr(), 7, 0, 100, None, None, None, 3)
    for name, hint in (('bpl4', capi.FLAG_TILE_LARGE), ('bpl1', capi.FLAG_TILE_SMALL)):
        times = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for k in range(4):
                lib.rollout(0, s, n, 0, b.data_ptr(), 7, 1000 + 64 * (4 * rep + k), T, None, None, None, 3 | hint)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3 / (4 * T))
        us = statistics.median(times)
        waves_per_simd = n / (64 * (4 if name == 'bpl4' else 1)) / 1024
        out[f'2^{lg} {name}'] = {'us_per_step': round(us, 3), 'us_per_2pow20_board_steps': round(us * (1 << 20) / n, 3),
                                 'waves_per_simd_offered': waves_per_simd}
print(json.dumps(out, indent=1))
