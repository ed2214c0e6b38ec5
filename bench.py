#!/usr/bin/env python3
"""
bench.py -- env board-steps/s of the MI355X-native batched 2048 environment.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch: one env2048_step launch over every board of
this rank's slice (slide/merge, reward, Philox spawn, terminal check, auto-reset), reading host-
provided uint8 actions from HBM.  Workload = BASELINE.json configs[2]: 1 048 576 boards per GPU
(weak scaling: every GPU owns a contiguous slice of 2^20 global environment indices, no collective
on the hot path), boards brought to mid-game by 64 random warm-up steps.

Timed region: barrier + torch.cuda.synchronize() | K launches | torch.cuda.synchronize() + barrier;
value = boards stepped by all ranks / max-over-ranks wall time.  The roofline object prices the
step kernel's ALGORITHMIC bytes (38 B per board-step) against its average launch
duration measured with HIP events on the launch stream over the same timed region.

Prints exactly one JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

ALGO_BYTES_PER_STEP = 38  # 16 board in + 1 action in + 16 board out + 4 reward out + 1 done out
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290
SEED = 0x2048


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--boards', type=int, default=1 << 20, help='boards per GPU')
    ap.add_argument('--mode', choices=['eager', 'graph'], default='eager',
                    help='eager: one C-ABI call per step; graph: the K launches captured once in a hipGraph')
    ap.add_argument('--chains', type=int, default=1, help='graph mode only: cut the slice into this many independent chains of '
                    'launches on separate streams inside the graph (experiment; 1 = one launch per step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target duration of the CPU baseline sample')
    ap.add_argument('--extras', action='store_true', help='also emit legal mask + obs (not the headline)')
    ap.add_argument('--obs-dtype', choices=['f32', 'f16', 'bf16'], default='f32', help='with --extras: element type of obs')
    ap.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL, one GPU per rank); 'gloo' only to rehearse the "
                    'multi-rank logic on a single-GPU box (all ranks then share device 0)')
    return ap.parse_args()


def cpu_baseline(boards_np, seconds: float):
    """Time the CPU oracle (C restatement of the reference engine, OpenMP) on the same workload.
    Checker infrastructure used as a *reported baseline* only -- never on the product path."""
    import numpy as np

    import oracle

    threads = usable_cores()
    if not oracle.openmp_enabled():
        threads = 1
    n = boards_np.shape[0]
    boards = boards_np.copy()
    rng = np.random.default_rng(1)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    reward = np.zeros(n, dtype=np.int32)
    done = np.zeros(n, dtype=np.uint8)
    oracle.step_batch_inplace(boards, actions, reward, done, SEED, 10_000, flags=1, threads=threads)  # warm
    steps = 0
    t0 = time.perf_counter()
    while True:
        oracle.step_batch_inplace(boards, actions, reward, done, SEED, 10_001 + steps, flags=1, threads=threads)
        steps += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= seconds or steps >= 100_000:
            break
    # the same loop on ONE thread for ~2 s, quoted in the sample text (per-core figure next to the reference's)
    single_steps, t1 = 0, time.perf_counter()
    while threads > 1:
        oracle.step_batch_inplace(boards, actions, reward, done, SEED, 500_000 + single_steps, flags=1, threads=1)
        single_steps += 1
        single_elapsed = time.perf_counter() - t1
        if single_elapsed >= 2.0:
            break
    single = f'; 1 thread: {n * single_steps / single_elapsed:.2e} steps/s' if single_steps else ''
    return {
        'value': n * steps / elapsed,
        'unit': 'board-steps/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'{steps} steps x {n} boards (same mid-game boards, auto-reset, host uint8 actions) in {elapsed:.1f} s{single}; '
                  f'oracle/env2048_oracle.c, gcc -O2 -fopenmp; reference NumPy engine itself: ~1.0e4 steps/s/core (BASELINE.md)',
    }


def usable_cores() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota and by
    the 16-core share a one-GPU box gets (an oversubscribed OpenMP team is slower, not faster)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = Path('/sys/fs/cgroup/cpu.max').read_text().split()
        if quota != 'max':
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(cores, int(os.environ.get('ENV2048_CPU_THREADS', '16'))))


def load_traffic(boards: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    path = REPO / 'profiles' / 'traffic.json'
    if not path.exists():
        return None
    try:
        table = json.loads(path.read_text())
        entry = table.get(str(boards))
        return None if entry is None else entry.get('hbm_bytes_per_launch')
    except Exception:
        return None


def main():
    args = parse_args()
    import torch

    from simulate_2048_amd import capi
    from simulate_2048_amd.batch import BatchedEnv2048

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run --nproc-per-node {args.gpus}')
        sys.exit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a HIP device: the product path has no CPU fallback')
    rehearsal = world > 1 and args.dist_backend != 'nccl'
    device_index = 0 if rehearsal else local_rank
    if device_index >= torch.cuda.device_count():
        sys.exit(f'rank {rank}: local rank {local_rank} has no GPU (visible devices: {torch.cuda.device_count()})')
    torch.cuda.set_device(device_index)
    device = torch.device('cuda', device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    n = args.boards
    K, W = args.steps, args.warmup
    obs_bytes = 4 if args.obs_dtype == 'f32' else 2
    obs_dtype = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[args.obs_dtype]
    env = BatchedEnv2048(n, SEED, device=device, env_base=rank * n, auto_reset=True, with_legal=args.extras,
                         with_obs=args.extras, obs_dtype=obs_dtype)
    env.reset()
    for _ in range(64):  # bring the boards to mid-game
        env.step(None)
    torch.cuda.synchronize()
    boards_midgame = env.boards.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    # host-provided actions: a small pool of uint8 arrays, cycled
    gen = torch.Generator(device='cpu').manual_seed(1 + rank)
    pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).to(device) for _ in range(8)]

    lib = capi.library()
    step_fn = lib._dll.env2048_step
    stream = torch.cuda.current_stream(device)
    fixed = dict(
        boards=ctypes.c_void_p(env.boards.data_ptr()), reward=ctypes.c_void_p(env.reward.data_ptr()),
        done=ctypes.c_void_p(env.done.data_ptr()),
        legal=ctypes.c_void_p(env.legal.data_ptr()) if env.legal is not None else None,
        obs=ctypes.c_void_p(env.obs.data_ptr()) if env.obs is not None else None,
    )
    action_ptrs = [ctypes.c_void_p(a.data_ptr()) for a in pool]
    env_base = rank * n

    def launch(step_index: int, stream_handle):
        rc = step_fn(device_index, stream_handle, n, env_base, fixed['boards'], action_ptrs[step_index & 7], fixed['reward'],
                     fixed['done'], fixed['legal'], fixed['obs'], SEED, step_index, env.flags)
        if rc != 0:
            raise capi.Env2048Error(rc, lib.last_error())

    def launch_part(step_index: int, stream_handle, part: int, parts: int):
        lo = part * (n // parts)
        cnt = n // parts
        off = lambda p, scale: None if p is None else ctypes.c_void_p(p.value + lo * scale)
        rc = step_fn(device_index, stream_handle, cnt, env_base + lo, off(fixed['boards'], 16), off(action_ptrs[step_index & 7], 1),
                     off(fixed['reward'], 4), off(fixed['done'], 1), off(fixed['legal'], 1), off(fixed['obs'], 16 * obs_bytes), SEED,
                     step_index, env.flags)
        if rc != 0:
            raise capi.Env2048Error(rc, lib.last_error())

    def barrier():
        if dist is not None:
            dist.barrier()

    base_step = env.step_index
    graph = None
    if args.mode == 'graph':
        side = torch.cuda.Stream(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            for t in range(3):
                launch(base_step + t, side.cuda_stream)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                cap_stream = torch.cuda.current_stream(device)
                if args.chains <= 1:
                    for t in range(K):
                        launch(base_step + W + t, cap_stream.cuda_stream)
                else:
                    others = [torch.cuda.Stream(device) for _ in range(args.chains - 1)]
                    fork = torch.cuda.Event()
                    fork.record(cap_stream)
                    for o in others:
                        o.wait_event(fork)
                    streams = [cap_stream] + others
                    for t in range(K):
                        for c, st in enumerate(streams):
                            launch_part(base_step + W + t, st.cuda_stream, c, args.chains)
                    for o in others:
                        ev = torch.cuda.Event()
                        ev.record(o)
                        cap_stream.wait_event(ev)
        torch.cuda.synchronize()

    h = stream.cuda_stream
    for t in range(W):
        launch(base_step + t, h)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        for t in range(K):
            launch(base_step + W + t, h)
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    event_ms = ev0.elapsed_time(ev1)

    if dist is not None:
        tt = torch.tensor([wall, event_ms], dtype=torch.float64, device=device if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, event_ms = float(tt[0]), float(tt[1])

    if rank == 0:
        total_steps = float(n) * K * world
        value = total_steps / wall
        launch_s = event_ms * 1e-3 / K
        bytes_per_launch = ALGO_BYTES_PER_STEP * n + (n * (1 + 16 * obs_bytes) if args.extras else 0)
        achieved = bytes_per_launch / launch_s / 1e9
        out = {
            'metric': 'env_board_steps_per_sec',
            'value': value,
            'unit': 'board-steps/s',
            'n_gpus': world,
            'steps': K,
            'warmup': W,
            'ms_per_step': wall * 1e3 / K,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u8',
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[2]: 1 048 576 parallel 4x4 boards per GPU, uint8 log2-packed, uniform random '
                            'host actions, Philox spawn, terminal check, auto-reset; boards mid-game (64 warm-up steps)',
                'boards_per_gpu': n,
                'global_boards': n * world,
                'launch_mode': args.mode,
                'sharding': 'contiguous env-index slices, one process per GPU, no collective on the hot path',
                'outputs': 'board,reward,done' + (f',legal,obs({args.obs_dtype})' if args.extras else ''),
            },
            'roofline': {
                'bound': 'hbm',
                'achieved': achieved,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': load_traffic(n),
                'kernel': 'step_kernel<false, false>' if not args.extras else 'step_kernel<true, false>',
                'algorithmic_bytes_per_launch': bytes_per_launch,
                'avg_launch_us': launch_s * 1e6,
            },
        }
        if boards_midgame is not None:
            out['cpu_baseline'] = cpu_baseline(boards_midgame, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
