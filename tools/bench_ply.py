#!/usr/bin/env python3
"""Throughput + roofline of the fused expectimax ply (env2048_expectimax_ply) next to the unfused chain
env2048_afterstates4 -> env2048_chance_outcomes -> env2048_status, on mid-game boards.  Run on the GPU box:
    python3 tools/bench_ply.py [--boards 262144] > gpurun_out/ply.json"""
import argparse
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import BatchedEnv2048, capi  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps=30, warm=5):
    stream = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return statistics.median(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--boards', type=int, default=1 << 18)
    ap.add_argument('--lib', default=None, help='experiment build of libenv2048.so to time instead of the product library')
    args = ap.parse_args()
    n = args.boards
    env = BatchedEnv2048(n, 0x2048, device='cuda:0', auto_reset=True)
    env.reset()
    env.rollout(64)
    lib = capi.Library(args.lib) if args.lib else capi.library()
    dev, s = 0, torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    b = env.boards
    sb = torch.empty((n, 4, 32, 16), dtype=torch.uint8, device='cuda')
    sp = torch.empty((n, 4, 32), dtype=torch.float32, device='cuda')
    ss = torch.empty((n, 4, 32), dtype=torch.uint8, device='cuda')
    sc = torch.empty((n, 4), dtype=torch.uint8, device='cuda')
    ar = torch.empty((n, 4), dtype=torch.int32, device='cuda')
    lg = torch.empty((n,), dtype=torch.uint8, device='cuda')
    a4 = torch.empty((n, 4, 16), dtype=torch.uint8, device='cuda')
    st_l = torch.empty((n * 128,), dtype=torch.uint8, device='cuda')
    st_d = torch.empty((n * 128,), dtype=torch.uint8, device='cuda')

    full = timed(lambda: lib.expectimax_ply(dev, s, n, p(b), p(sb), p(sp), p(ss), p(sc), p(ar), p(lg)))
    lean = timed(lambda: lib.expectimax_ply(dev, s, n, p(b), p(sb), p(sp), None, None, None, None))

    def chain():
        lib.afterstates4(dev, s, n, p(b), p(a4), p(ar), p(lg))
        lib.chance_outcomes(dev, s, 4 * n, p(a4), p(sb), p(sp), None)
        lib.status(dev, s, 128 * n, p(sb), p(st_d), p(st_l), None, None, None)

    unfused = timed(chain)
    bytes_lean = n * (16 + 128 * (16 + 4))
    bytes_full = bytes_lean + n * (128 + 4 + 16 + 1)
    bytes_chain = n * (16 + 64 + 16 + 1) + n * (64 + 128 * 20) + n * 128 * (16 + 2)
    out = {
        'kernel': 'expectimax_ply_kernel',
        'boards': n,
        'successors_per_board': 128,
        'fused_with_status': {'us': full * 1e6, 'boards_per_s': n / full, 'successors_per_s': 128 * n / full,
                              'roofline': {'bound': 'hbm', 'achieved': bytes_full / full / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                           'frac': bytes_full / full / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': bytes_full,
                                           'bytes_per_board': bytes_full // n}},
        'fused_boards_and_probs_only': {'us': lean * 1e6, 'boards_per_s': n / lean,
                                        'roofline': {'bound': 'hbm', 'achieved': bytes_lean / lean / 1e9, 'peak': HBM_PEAK_GBS,
                                                     'unit': 'GB/s', 'frac': bytes_lean / lean / 1e9 / HBM_PEAK_GBS,
                                                     'algorithmic_bytes_per_launch': bytes_lean, 'bytes_per_board': bytes_lean // n}},
        'unfused_chain_3_launches': {'us': unfused * 1e6, 'boards_per_s': n / unfused, 'bytes_moved': bytes_chain,
                                     'achieved_GBps': bytes_chain / unfused / 1e9},
        'speedup_fused_vs_chain': unfused / full,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
