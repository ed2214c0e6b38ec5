#!/usr/bin/env python3
"""Every batch entry point of the C-ABI once, at 2^22 boards of mid-game play (HBM-bound sizes: the working sets exceed the
Infinity Cache for most of them): microseconds per launch, the algorithmic bytes the call moves, and their rate -- the table that
shows which kernel is how far from the memory system's ceiling (GPU box).  python3 tools/bench_all_entries.py [log2_boards]"""
import json
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

import os  # noqa: E402

lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library()  # (an A/B build)
S = torch.cuda.current_stream().cuda_stream
LG = int(sys.argv[1]) if len(sys.argv) > 1 else 22
N = 1 << LG
dev = 'cuda'


def timed(fn, reps=10):
    fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def z(shape, dtype):
    return torch.zeros(shape, dtype=dtype, device=dev)


p = lambda t: None if t is None else t.data_ptr()
boards = z((N, 16), torch.uint8)
lib.reset(0, S, N, 0, p(boards), None, 0x2048, 0)
lib.rollout(0, S, N, 0, p(boards), 0x2048, 0, 100, None, None, None, 3)
saved = boards.clone()
acts = torch.randint(0, 4, (N,), dtype=torch.uint8, device=dev)
acts32 = acts.to(torch.int32)
reward, done, legal, maxe, nempty = z((N,), torch.int32), z((N,), torch.uint8), z((N,), torch.uint8), z((N,), torch.uint8), z((N,), torch.uint8)
obs = z((N, 16), torch.float32)
out_boards = z((N, 16), torch.uint8)
changed = z((N,), torch.uint8)
vals32, vals64 = z((N, 16), torch.int32), z((N, 16), torch.int64)
lib.unpack_i32(0, S, N, p(boards), p(vals32))
lib.unpack_i64(0, S, N, p(boards), p(vals64))
out32 = z((N, 16), torch.int32)
legal4 = z((N, 4), torch.uint8)
keys = torch.randint(0, 1 << 31, (N, 2), dtype=torch.int32, device=dev).view(torch.int32)
sc, tot, rewf, mt = z((N,), torch.int32), z((N,), torch.float32), z((N,), torch.float32), z((N,), torch.int32)
rows = {}


def row(name, nbytes, fn, reps=10):
    us = timed(fn, reps)
    rows[name] = {'us_per_launch': round(us, 2), 'bytes_per_board': nbytes, 'TB_per_s': round(N * nbytes / us / 1e6, 3), 'frac_of_8TBps': round(N * nbytes / us / 8e6, 3)}
    print(name, rows[name], flush=True)


row('reset', 16 + 1, lambda: lib.reset(0, S, N, 0, p(out_boards), p(done), 7, 3))
row('step (host-chosen moves)', 38, lambda: lib.step(0, S, N, 0, p(boards), p(acts), p(reward), p(done), None, None, 7, 5, capi.FLAG_AUTO_RESET))
boards.copy_(saved)
row('step + legal + obs f32', 103, lambda: lib.step(0, S, N, 0, p(boards), p(acts), p(reward), p(done), p(legal), p(obs), 7, 5, capi.FLAG_AUTO_RESET))
boards.copy_(saved)
row('status (done, legal, max_exp, n_empty)', 16 + 4, lambda: lib.status(0, S, N, p(boards), p(done), p(legal), None, p(maxe), p(nempty)))
row('status + obs f32', 16 + 4 + 64, lambda: lib.status(0, S, N, p(boards), p(done), p(legal), p(obs), p(maxe), p(nempty)))
row('afterstate', 16 + 1 + 16 + 4 + 1, lambda: lib.afterstate(0, S, N, p(boards), p(acts), p(out_boards), p(reward), p(changed)))
a4_boards, a4_reward = z((N, 4, 16), torch.uint8), z((N, 4), torch.int32)
row('afterstates4', 97, lambda: lib.afterstates4(0, S, N, p(boards), p(a4_boards), p(a4_reward), p(legal)))
del a4_boards, a4_reward
row('spawn_random (1 tile)', 32, lambda: lib.spawn_random(0, S, N, 0, p(out_boards), 7, 9, 1))
row('random_actions', 1, lambda: lib.random_actions(0, S, N, 0, 7, 9, p(acts)))
row('pack_i32', 64 + 16, lambda: lib.pack_i32(0, S, N, p(vals32), p(out_boards)))
row('unpack_i32', 64 + 16, lambda: lib.unpack_i32(0, S, N, p(boards), p(out32)))
row('pack_i64', 128 + 16, lambda: lib.pack_i64(0, S, N, p(vals64), p(out_boards)))
row('unpack_i64', 128 + 16, lambda: lib.unpack_i64(0, S, N, p(boards), p(vals64)))
row('status_i32 (obs, legal4, done)', 64 + 64 + 4 + 1, lambda: lib.status_i32(0, S, N, p(vals32), p(obs), p(legal4), p(done)))
row('afterstate_i32', 64 + 4 + 64 + 4 + 1, lambda: lib.afterstate_i32(0, S, N, p(vals32), p(acts32), 0, p(out32), p(reward), p(changed)))
row('step_state_i32_keyed_status (the batched mirror move)', 8 + 64 + 4 + 4 + 1 + 4 + 64 + 4 + 1 + 4 + 4 + 4 + 64 + 4,
    lambda: lib.step_state_i32_keyed_status(0, S, N, p(keys), p(vals32), p(out32), p(acts32), p(sc), p(sc), p(done), p(changed), p(tot), p(tot), p(rewf), p(mt),
                                            p(obs), p(legal4)))
row('reset_keyed_i32_status', 8 + 64 + 64 + 4, lambda: lib.reset_keyed_i32_status(0, S, N, p(keys), p(out32), p(obs), p(legal4)))
oh = z((N, 16, 31), torch.uint8) if N <= (1 << 22) else None
if oh is not None:
    row('encode_onehot (width 31)', 16 + 16 * 31, lambda: lib.encode_onehot(0, S, N, p(boards), 31, p(oh)))
    del oh
M = N >> 2  # the enumerations write hundreds of bytes per board
co_b, co_p, co_c = z((M, 32, 16), torch.uint8), z((M, 32), torch.float32), z((M,), torch.uint8)
us = timed(lambda: lib.chance_outcomes(0, S, M, p(boards), p(co_b), p(co_p), p(co_c)))
rows['chance_outcomes (N/4 boards)'] = {'us_per_launch': round(us, 2), 'bytes_per_board': 657, 'TB_per_s': round(M * 657 / us / 1e6, 3), 'frac_of_8TBps': round(M * 657 / us / 8e6, 3)}
print('chance_outcomes', rows['chance_outcomes (N/4 boards)'], flush=True)
del co_b, co_p, co_c
for size in (6, 8):
    cells = size * size
    nb = z((N, cells), torch.uint8)
    lib.nxn_reset(0, S, N, size, 0, p(nb), None, 1, 0)
    for k in range(60):
        lib.nxn_step(0, S, N, size, 0, p(nb), None, p(reward), p(done), None, 1, k + 1, 3)
    nout = z((N, cells), torch.uint8)
    row(f'nxn_afterstate S={size}', 2 * cells + 6, lambda: lib.nxn_afterstate(0, S, N, size, p(nb), p(acts), p(nout), p(reward), p(changed)), reps=5)
    row(f'nxn_status S={size}', cells + 4, lambda: lib.nxn_status(0, S, N, size, p(nb), p(done), p(legal), p(nempty), p(maxe)), reps=5)
    row(f'nxn_reset S={size}', cells + 1, lambda: lib.nxn_reset(0, S, N, size, 0, p(nout), p(done), 1, 0), reps=5)
    row(f'nxn_spawn_random S={size}', 2 * cells, lambda: lib.nxn_spawn_random(0, S, N, size, 0, p(nout), 1, 5, 1), reps=5)
    row(f'nxn_step S={size}', 2 * cells + 5, lambda: lib.nxn_step(0, S, N, size, 0, p(nb), None, p(reward), p(done), None, 1, 99, 3), reps=5)
    del nb, nout
print(json.dumps({'boards': N, 'rows': rows}, indent=1))
