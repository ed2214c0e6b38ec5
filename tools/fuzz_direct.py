"""Randomised differential run of the direct route of env2048_step_sequence (csrc/direct_dispatch.hpp) on the GPU box: random batch
sizes (1 ... 1.5 x 2^20 and just beyond, ragged ends, both tilings, one and two chains), sequence lengths 8 ... 700 (several argument-ring segments),
host action rows or in-kernel moves, auto-reset on / off, outputs in place / one row per step / padded rows, reward present or
NULL, misaligned-by-16 board bases, busy caller streams -- every call compared with the same call on the stream route
(ENV2048_FLAG_NO_DIRECT) byte for byte, small ones also with the oracle step by step.
python tools/fuzz_direct.py [seconds] [--churn]   (--churn: a background thread creates and destroys queues in THIS process all the
time, tools/queue_churn.c -- the XCD placement of the library's queues then moves while sequences run and some must be re-run)"""
import ctypes
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import oracle
from simulate_2048_amd import capi

lib = capi.library()
info = lib.direct_stats(0)
assert info['available'], info['reason']
rng = np.random.default_rng(int(time.time()))
budget = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 60.0
if '--churn' in sys.argv:
    import threading

    import subprocess

    churn_lib = Path(__file__).resolve().parent / 'libqueue_churn.so'
    if not churn_lib.exists():
        subprocess.run(['gcc', '-O2', '-shared', '-fPIC', str(churn_lib.with_name('queue_churn.c')), '-I/opt/rocm/include', '-L/opt/rocm/lib', '-lhsa-runtime64',
                        '-o', str(churn_lib)], check=True)
    churn = ctypes.CDLL(str(churn_lib))
    threading.Thread(target=lambda: churn.queue_churn(1_000_000, 700), daemon=True).start()
t0 = last = time.time()
counts = {'extras_calls': 0, 'calls': 0, 'steps': 0, 'board_steps': 0, 'oracle_checked': 0, 'two_chain_calls': 0, 'busy_stream_calls': 0}
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
AUTO, RANDOM, NO_DIRECT = capi.FLAG_AUTO_RESET, capi.FLAG_RANDOM_ACTIONS, capi.FLAG_NO_DIRECT
ballast = torch.empty((1 << 27,), dtype=torch.uint8, device='cuda')


def pick_n():
    kind = rng.integers(0, 6)
    if kind == 0:
        return int(rng.integers(1, 5000))
    if kind == 1:
        return int(rng.choice([511, 512, 513, 2047, 2048, 2049, 65536, (1 << 18) - 1, 1 << 18, (1 << 18) + 1]))
    if kind == 2:
        return int(rng.integers(1 << 18, 1 << 19))        # two chains, halves in the small tiling
    if kind == 3:
        return int(rng.integers(1 << 19, (1 << 20) + 5000))  # two chains, large tiling
    if kind == 4:
        return int(rng.choice([1 << 20, (1 << 20) + 2048, (3 << 19) - 1, 3 << 19, (3 << 19) + 1]))
    return int(rng.integers(5000, 1 << 18))


while time.time() - t0 < budget:
    n = pick_n()
    K = int(rng.choice([8, 9, 16, 20, 33, 64, 255, 256, 257, 300, 700])) if n < (1 << 19) else int(rng.choice([8, 12, 20, 40]))
    if n * K > 3e8:
        K = max(8, int(3e8 // n))
    flags = (AUTO if rng.random() < 0.7 else 0) | (RANDOM if rng.random() < 0.4 else 0)
    base = int(rng.integers(0, 1 << 40))
    seed = int(rng.integers(0, 1 << 63))
    step0 = int(rng.integers(0, 1 << 40))
    stride = int(rng.choice([0, n, n + int(rng.integers(1, 300))]))
    if stride * K > 6e8:
        stride = 0
    rows = K if stride else 1
    with_reward = rng.random() < 0.85
    extras = rng.random() < 0.3 and n * K <= 6e7   # legal mask + observation rows (f32 / f16 / bf16), sometimes the normalised reward
    obs_flag, obs_dtype = [(0, torch.float32), (capi.FLAG_OBS_F16, torch.float16), (capi.FLAG_OBS_BF16, torch.bfloat16)][int(rng.integers(0, 3))]
    normalised = extras and with_reward and rng.random() < 0.3
    with_obs = extras and rng.random() < 0.8
    if extras:
        flags |= obs_flag | (capi.FLAG_REWARD_LOG2_F32 if normalised else 0)
    shift = int(rng.integers(0, 4))  # board base 16-byte aligned but not line aligned
    start = torch.zeros((n + 8, 16), dtype=torch.uint8, device='cuda')[shift:shift + n]
    lib.reset(0, S(), n, base, P(start), None, seed ^ 5, 0)
    if rng.random() < 0.8:
        lib.rollout(0, S(), n, base, P(start), seed ^ 5, 0, int(rng.integers(1, 120)), None, None, None, AUTO | RANDOM)
    acts = None
    if not flags & RANDOM:
        pool = [torch.from_numpy(rng.integers(0, 256, size=n).astype(np.uint8)).cuda() for _ in range(min(K, 5))]
        acts = [pool[int(rng.integers(0, len(pool)))] for _ in range(K)]
    table = None if acts is None else (ctypes.c_void_p * K)(*[a.data_ptr() for a in acts])
    torch.cuda.synchronize()
    out = []
    busy = rng.random() < 0.3
    for extra in (0, NO_DIRECT):
        b = torch.zeros((n + 8, 16), dtype=torch.uint8, device='cuda')[shift:shift + n]
        b.copy_(start)
        r = torch.full((rows, max(stride, n)), -3, dtype=torch.float32 if normalised else torch.int32, device='cuda') if with_reward else None
        d = torch.full((rows, max(stride, n)), 7, dtype=torch.uint8, device='cuda')
        lg = torch.full((rows, max(stride, n)), 9, dtype=torch.uint8, device='cuda') if extras else None
        ob = torch.full((rows, max(stride, n), 16), -2.0, dtype=obs_dtype, device='cuda') if with_obs else None
        before = lib.direct_stats(0)['calls']
        if busy and extra == 0:
            ballast.fill_(3)  # the caller's stream is busy when the call arrives
        lib.step_sequence(0, S(), n, base, P(b), table, P(r), P(d), P(lg), P(ob), stride, seed, step0, K, flags | extra)
        assert lib.direct_stats(0)['calls'] == before + (1 if extra == 0 and lib.direct_eligible(n, K, flags) else 0), 'route taken'
        out.append((b, r, d, lg, ob))
    torch.cuda.synchronize()
    (b1, r1, d1, l1, o1), (b2, r2, d2, l2, o2) = out
    assert torch.equal(b1, b2) and torch.equal(d1, d2) and (r1 is None or torch.equal(r1, r2)), ('direct != stream', n, K, flags, stride)
    assert (l1 is None or torch.equal(l1, l2)) and (o1 is None or (o2 is not None and torch.equal(o1, o2))), ('direct != stream (legal / obs)', n, K, flags, stride)
    counts['extras_calls'] += int(extras)
    if n * K <= 3e6 and acts is not None and not extras:
        ref = start.cpu().numpy().copy()
        for t in range(K):
            want = oracle.step_batch(ref, acts[t].cpu().numpy(), seed, step0 + t, env_base=base, flags=flags)
            ref = want['boards']
        assert np.array_equal(b1.cpu().numpy(), ref), ('direct != oracle', n, K)
        counts['oracle_checked'] += 1
    counts['calls'] += 1
    counts['steps'] += K
    counts['board_steps'] += n * K
    counts['two_chain_calls'] += int(n >= 1 << 18)
    counts['busy_stream_calls'] += int(busy)
    if time.time() - last > 30:
        last = time.time()
        print('fuzz_direct:', counts, lib.direct_stats(0)['reruns'], flush=True)
final = lib.direct_stats(0)
print('fuzz_direct done, all identical:', counts, 'reruns', final['reruns'], 'available', final['available'])
