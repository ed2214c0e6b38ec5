#!/usr/bin/env python3
"""Probe: does buffer placement (torch caching allocator vs raw hipMalloc) change step-kernel time?"""
import ctypes
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from simulate_2048_amd import capi
from simulate_2048_amd.batch import BatchedEnv2048

hip = ctypes.CDLL('libamdhip64.so')
n = 1 << 20
env = BatchedEnv2048(n, 0x2048, device='cuda:0', auto_reset=True)
env.reset()
for _ in range(64):
    env.step(None)
torch.cuda.synchronize()
lib = capi.library()
fn = lib._dll.env2048_step
stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator().manual_seed(1)
pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).cuda() for _ in range(8)]


def hipmalloc(nbytes):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
    return p


def timed(boards, acts, reward, done, label, K=1500):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for t in range(200):
        fn(0, stream, n, 0, boards, acts[t & 7], reward, done, None, None, 0x2048, 1000 + t, 1)
    torch.cuda.synchronize()
    e0.record()
    for t in range(K):
        fn(0, stream, n, 0, boards, acts[t & 7], reward, done, None, None, 0x2048, 2000 + t, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f'{label:40s} {e0.elapsed_time(e1) * 1e3 / K:7.2f} us/launch', flush=True)


P = ctypes.c_void_p
timed(P(env.boards.data_ptr()), [P(a.data_ptr()) for a in pool], P(env.reward.data_ptr()), P(env.done.data_ptr()), 'torch tensors')
# raw hipMalloc buffers with the same contents
rb, rr, rd = hipmalloc(n * 16), hipmalloc(n * 4), hipmalloc(n)
ra = [hipmalloc(n) for _ in range(8)]
hip.hipMemcpy(rb, P(env.boards.data_ptr()), ctypes.c_size_t(n * 16), 3)
for dst, src in zip(ra, pool):
    hip.hipMemcpy(dst, P(src.data_ptr()), ctypes.c_size_t(n), 3)
timed(rb, ra, rr, rd, 'raw hipMalloc buffers')
# torch tensors again but with ONE contiguous action array (like the C++ harness)
acts = torch.cat(pool)
timed(P(env.boards.data_ptr()), [P(acts.data_ptr() + i * n) for i in range(8)], P(env.reward.data_ptr()), P(env.done.data_ptr()),
      'torch tensors, contiguous action pool')
# hash-pattern actions like the C++ harness
idx = torch.arange(8 * n, dtype=torch.int64)
pat = (((idx * 2654435761) >> 13) & 3).to(torch.uint8).cuda()
timed(P(env.boards.data_ptr()), [P(pat.data_ptr() + i * n) for i in range(8)], P(env.reward.data_ptr()), P(env.done.data_ptr()),
      'torch tensors, C++-style action pattern')
big = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')  # separate 64 MiB segment, explicit offsets
timed(P(big.data_ptr()), [P(big.data_ptr() + (32 << 20) + i * n) for i in range(8)], P(big.data_ptr() + (20 << 20)),
      P(big.data_ptr() + (28 << 20)), 'one 64MiB torch segment (garbage data)')
