#!/usr/bin/env python3
"""Launch-bound regime (small batches): K dependent env2048_step launches enqueued eagerly (env2048_step_sequence) vs the
same K launches captured once in a hipGraph and replayed -- per-step time on the GPU by HIP events."""
import ctypes
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

SEED = 0x2048


def main():
    lib = capi.Library(sys.argv[1]) if len(sys.argv) > 1 else capi.library()
    res = {}
    K = 1000
    for n in (8, 1 << 12, 1 << 16, 1 << 18, 1 << 20):
        b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
        r = torch.zeros((n,), dtype=torch.int32, device='cuda')
        d = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        s0 = torch.cuda.current_stream().cuda_stream
        lib.reset(0, s0, n, 0, b.data_ptr(), d.data_ptr(), SEED, 0)
        lib.rollout(0, s0, n, 0, b.data_ptr(), SEED, 0, 64, None, None, None, 3)
        gen = torch.Generator(device='cpu').manual_seed(7)
        pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]
        table = (ctypes.c_void_p * K)(*[pool[t & 7].data_ptr() for t in range(K)])
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            lib.step_sequence(0, side.cuda_stream, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, SEED, 64, 8, 1)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                cap = torch.cuda.current_stream().cuda_stream
                lib.step_sequence(0, cap, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, SEED, 100, K, 1)
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream()

        def timed(fn):
            out = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(stream)
                fn()
                e1.record(stream)
                torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) * 1e3 / K)
            return round(statistics.median(out), 3)

        eager = timed(lambda: lib.step_sequence(0, stream.cuda_stream, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0,
                                                SEED, 100, K, 1))
        replay = timed(graph.replay)
        res[n] = {'eager_us_per_step': eager, 'graph_replay_us_per_step': replay}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
