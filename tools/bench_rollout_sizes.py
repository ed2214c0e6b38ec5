#!/usr/bin/env python3
"""Fused rollout (512 steps) over batch sizes, both tilings: where one board per lane beats four."""
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

lib = capi.library()
s = torch.cuda.current_stream().cuda_stream
stream = torch.cuda.current_stream()
out = {}
for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 1 << 20):
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, s, n, 0, b.data_ptr(), None, 1, 0)
    row = {}
    for name, tile in (('small', capi.FLAG_TILE_SMALL), ('large', capi.FLAG_TILE_LARGE)):
        times = []
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            lib.rollout(0, s, n, 0, b.data_ptr(), 1, rep * 512, 512, None, None, None, 3 | tile)
            e1.record(stream)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        row[name + '_ms'] = round(statistics.median(times), 4)
    out[n] = row
print(json.dumps(out, indent=1))
