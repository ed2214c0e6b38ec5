#!/usr/bin/env python3
"""Where the time of ops.expectimax(depth 3, 2^17 roots) goes (GPU box): every phase timed with HIP events on the stream and
with the host clock (each phase synchronised -- a diagnosis, not the product path)."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi, ops, runtime  # noqa: E402


def main():
    import os

    lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library()  # (an experiment build: tools/build_exp.sh)
    rt = runtime.Runtime(lib, 'cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    n = 1 << 17
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, s, n, 0, b.data_ptr(), None, 0x2048, 0)
    lib.rollout(0, s, n, 0, b.data_ptr(), 0x2048, 0, 100, None, None, None, 3)
    out = {}

    def timed(name, fn, reps=7):
        fn()
        torch.cuda.synchronize()
        ev, wall = [], []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            r = fn()
            e1.record()
            torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            ev.append(e0.elapsed_time(e1) * 1e-3)
        out[name] = {'event_us': round(sorted(ev)[reps // 2] * 1e6, 1), 'wall_us': round(sorted(wall)[reps // 2] * 1e6, 1)}
        return r

    counts = torch.empty((n,), dtype=torch.int32, device='cuda')
    timed('expand_count', lambda: lib.expand_count(0, s, n, b.data_ptr(), counts.data_ptr()))
    ends = timed('cumsum', lambda: torch.cumsum(counts, dim=0, dtype=torch.int64))
    offsets = timed('offsets_sub', lambda: ends - counts)
    total = timed('total_readback', lambda: int(ends[-1]))
    children = torch.empty((total, 16), dtype=torch.uint8, device='cuda')
    reward = torch.empty((n, 4), dtype=torch.int32, device='cuda')
    n_empty = torch.empty((n, 4), dtype=torch.uint8, device='cuda')
    timed('allocs', lambda: (torch.empty((total, 16), dtype=torch.uint8, device='cuda'), torch.empty((n, 4), dtype=torch.int32, device='cuda'),
                             torch.empty((total,), dtype=torch.float64, device='cuda')))
    timed('expand', lambda: lib.expand(0, s, n, b.data_ptr(), offsets.data_ptr(), children.data_ptr(), reward.data_ptr(), n_empty.data_ptr()))
    value = torch.empty((total,), dtype=torch.float64, device='cuda')
    timed('depth2_f64_leaves', lambda: lib.expectimax_depth2_f64(0, s, total, children.data_ptr(), value.data_ptr()))
    parent = torch.empty((n,), dtype=torch.float64, device='cuda')
    q = torch.empty((n, 4), dtype=torch.float64, device='cuda')
    best = torch.empty((n,), dtype=torch.uint8, device='cuda')
    timed('backup', lambda: lib.expectimax_backup(0, s, n, reward.data_ptr(), n_empty.data_ptr(), offsets.data_ptr(), value.data_ptr(), parent.data_ptr(),
                                                   q.data_ptr(), best.data_ptr()))
    timed('whole_expectimax_depth3', lambda: ops.expectimax(b, 3, rt=rt))
    out['leaves'] = total
    out['bytes'] = {'children': total * 16, 'child_values': total * 8}
    for name, nbytes in (('expand', total * 16 + n * 36), ('backup', total * 8 + n * 70), ('depth2_f64_leaves', total * 24)):
        out[name]['GBps'] = round(nbytes / out[name]['event_us'] / 1e3, 1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
