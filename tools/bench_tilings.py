#!/usr/bin/env python3
"""Per-step time of env2048_step in both tilings (ENV2048_FLAG_TILE_SMALL / _LARGE) over launch sizes: where the switch belongs."""
import ctypes
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

SEED = 0x2048


def time_case(lib, n, flags, count=1500):
    dev, s = 0, torch.cuda.current_stream().cuda_stream
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    r = torch.zeros((n,), dtype=torch.int32, device='cuda')
    d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
    lib.reset(dev, s, n, 0, b.data_ptr(), d.data_ptr(), SEED, 0)
    lib.rollout(dev, s, n, 0, b.data_ptr(), SEED, 0, 64, None, None, None, 3)
    gen = torch.Generator(device='cpu').manual_seed(7)
    pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]
    table = (ctypes.c_void_p * count)(*[pool[t & 7].data_ptr() for t in range(count)])
    stream = torch.cuda.current_stream()
    out = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        lib.step_sequence(dev, s, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, SEED, 64 + rep * count, count, 1 | flags)
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / count)
    return statistics.median(out)


def main():
    lib = capi.library()
    res = {}
    for n in (8, 1024, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20, 1 << 21):
        res[n] = {'small_1_board_per_lane': round(time_case(lib, n, capi.FLAG_TILE_SMALL), 3),
                  'large_4_boards_per_lane': round(time_case(lib, n, capi.FLAG_TILE_LARGE), 3)}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
