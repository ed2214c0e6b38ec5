#!/usr/bin/env python3
"""Where the microseconds of ONE reinforce.game move at N = 8 go (GPU box): raw C-ABI enqueue cost, launch -> completion-word
latency, and the Python pieces of batched_step (tools/bench_mirror_move.py gives the totals)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi, resultpool, runtime  # noqa: E402
from simulate_2048_amd.reinforce.game import env as genv  # noqa: E402


def main():
    n = 8
    out = {}
    rt = runtime.current()
    lib = rt.lib
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    acts = rng.integers(0, 4, size=n).astype(np.int32)
    states = genv.batched_reset(keys)
    a = resultpool.pool.lookup(states.board)[0]
    states2, *_ = genv.batched_step(states, acts, keys)
    b = resultpool.pool.lookup(states2.board)[0]
    torch.cuda.synchronize()
    dev, stream = rt.handles()
    kd = torch.from_numpy(keys.view(np.int32)).cuda()
    ad = torch.from_numpy(acts).cuda()
    kp, ap = kd.data_ptr(), ad.data_ptr()
    kpin = torch.from_numpy(keys.view(np.int32)).pin_memory()
    apin = torch.from_numpy(acts).pin_memory()

    def launch(src, dst, serial, kptr=kp, aptr=ap):
        p, q = src.ptr, dst.ptr
        lib.step_state_i32_keyed_signalled(dev, stream, n, kptr, p['board'], q['board'], aptr, p['step_count'], q['step_count'], p['done'],
                                           q['done'], p['total'], q['total'], q['reward'], q['max_tile'], q['obs'], q['legal'], dst.mirror_delta,
                                           dst.seq_ptr, serial)

    serial = 1 << 40
    reps = 3000
    # 1. enqueue only (the stream runs behind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        serial += 1
        launch(a, b, serial)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out['capi_enqueue_only_us'] = (t1 - t0) / reps * 1e6
    # 2. launch + wait for the completion word, inputs in device memory / in pinned host memory
    for label, kptr, aptr in (('device_inputs', kp, ap), ('pinned_host_inputs', kpin.data_ptr(), apin.data_ptr())):
        lat = []
        torch.cuda.synchronize()
        for r in range(reps):
            serial += 1
            t0 = time.perf_counter()
            launch(a, b, serial, kptr, aptr)
            t1 = time.perf_counter()
            ok = lib.wait_sequence(b.seq_ptr, serial, 1000)
            t2 = time.perf_counter()
            if not ok:
                torch.cuda.synchronize()
                word = int(b.mirror.numpy()[-16:-8].view(np.uint64)[0])
                raise AssertionError(f'{label} rep {r}: completion word {word:#x}, expected {serial:#x} (after a synchronize)')
            lat.append((t1 - t0, t2 - t1))
        lat = np.array(lat[reps // 10:]) * 1e6
        out[f'launch_call_us_{label}'] = float(np.median(lat[:, 0]))
        out[f'wait_after_launch_us_{label}'] = float(np.median(lat[:, 1]))
        out[f'wait_after_launch_p90_us_{label}'] = float(np.percentile(lat[:, 1], 90))
    # 3. same with stream.synchronize instead of the word
    lat = []
    s = torch.cuda.current_stream()
    for r in range(reps):
        serial += 1
        launch(a, b, serial)
        t1 = time.perf_counter()
        s.synchronize()
        lat.append(time.perf_counter() - t1)
    out['stream_synchronize_after_launch_us'] = float(np.median(lat) * 1e6)
    # 3b. a generic completion word: any launch + env2048_stream_signal (hipStreamWriteValue64) + poll
    word = torch.zeros((2,), dtype=torch.int64).pin_memory()
    lat = []
    ok_all = True
    for r in range(reps):
        serial += 1
        t0 = time.perf_counter()
        lib.step_state_i32_keyed_mirrored(dev, stream, n, kp, a.ptr['board'], b.ptr['board'], ap, a.ptr['step_count'], b.ptr['step_count'], a.ptr['done'], b.ptr['done'], a.ptr['total'], b.ptr['total'], b.ptr['reward'], b.ptr['max_tile'], b.ptr['obs'], b.ptr['legal'], b.mirror_delta)
        t1 = time.perf_counter()
        lib.stream_signal(dev, stream, word.data_ptr(), serial)
        t2 = time.perf_counter()
        ok_all &= lib.wait_sequence(word.data_ptr(), serial, 2000)
        lat.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
    lat = np.array(lat[reps // 10:]) * 1e6
    out['generic_signal'] = {'launch_call_us': float(np.median(lat[:, 0])), 'stream_signal_call_us': float(np.median(lat[:, 1])),
                             'wait_us': float(np.median(lat[:, 2])), 'all_seen': bool(ok_all)}
    # 4. an empty ctypes call / the wait call alone
    t0 = time.perf_counter()
    for r in range(reps):
        lib.wait_sequence(b.seq_ptr, serial, 0)
    out['wait_sequence_call_when_ready_us'] = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for r in range(reps):
        lib.abi_version()
    out['ctypes_noarg_call_us'] = (time.perf_counter() - t0) / reps * 1e6
    # 5. python pieces of batched_step
    def timeit(fn, reps=5000):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6
    out['py_state_ptrs_us'] = timeit(lambda: genv._state_ptrs(states, rt))
    out['py_keys_ptr_us'] = timeit(lambda: genv._keys_ptr(keys, rt))
    out['py_actions_ptr_us'] = timeit(lambda: genv._actions_ptr(acts, n, rt))
    out['py_handles_us'] = timeit(lambda: rt.handles())
    out['py_runtime_current_us'] = timeit(lambda: runtime.current())
    def acq():
        rs = resultpool.pool.acquire(n, rt.device, stream or 0)
        rs.leased = False
    out['py_pool_acquire_us'] = timeit(acq)
    out['py_np_array_of_ready_field_us'] = timeit(lambda: np.array(states2.board))
    out['py_get_observation_us'] = timeit(lambda: genv.batched_get_observation(states2))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
