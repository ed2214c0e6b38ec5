#!/usr/bin/env python3
"""Time ops.expectimax at depth 3 and 4 on mid-game boards (GPU box): expansions into dense frontiers, the depth-2 kernel on the
deepest frontier, back-ups.  Reports the frontier sizes, microseconds per phase and leaves (depth-2 evaluations) per second."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi, ops, runtime  # noqa: E402


def main():
    lib = capi.library()
    rt = runtime.Runtime(lib, 'cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for depth, lg in ((3, 10), (3, 14), (3, 17), (4, 8), (4, 12)):
        n = 1 << lg
        b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
        lib.reset(0, s, n, 0, b.data_ptr(), None, 0x2048, 0)
        lib.rollout(0, s, n, 0, b.data_ptr(), 0x2048, 0, 100, None, None, None, 3)
        for _ in range(2):
            res = ops.expectimax(b, depth, rt=rt)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            res = ops.expectimax(b, depth, rt=rt)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[2]
        out[f'depth{depth}_2^{lg}_roots'] = {'ms': round(dt * 1e3, 3), 'frontier': res['frontier'], 'roots_per_s': round(n / dt),
                                             'depth2_leaf_boards_per_s': round(res['frontier'][-1] / dt)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
