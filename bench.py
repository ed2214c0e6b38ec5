#!/usr/bin/env python3
"""
bench.py -- env board-steps/s of the MI355X-native batched 2048 environment.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script starts N fresh worker processes itself (one per
GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1) and relays rank 0's JSON
line; the parent never touches a GPU.  It can also be started the classic way:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch: one env2048_step launch over every board of
this rank's slice (slide/merge, reward, Philox spawn, terminal check, auto-reset), reading host-
provided uint8 actions from HBM.  Workload = BASELINE.json configs[2]: 2^20 boards per GPU (weak
scaling: every GPU owns a contiguous slice of global environment indices, no collective on the hot
path), boards brought to mid-game by 64 random warm-up steps.

Timed region (one "block"): barrier + torch.cuda.synchronize() | K launches enqueued from C by ONE
env2048_step_sequence call | torch.cuda.synchronize().  A block of K = 20 steps lasts 0.2 ms, so the
block is repeated R times (>= --min-ms of GPU time in total); every block is timed on its own (wall
clock and HIP events on the launch stream), the per-block MAX over ranks is taken, and the MEDIAN
block gives `value` / `ms_per_step` (min and spread are reported next to it).

The roofline object is the metric BASELINE.json asks for: bound "hbm", achieved =
ALGORITHMIC bytes per launch (38 B x boards) / the launch duration measured with HIP events in this
run, peak = 8 TB/s, frac = achieved / peak.  At 2^20 boards the working set (16 MiB boards + action
pool + outputs) lives in the 256 MiB Infinity Cache, so for the headline leg that fraction is a
cache-resident equivalent and says so (`regime`); `roofline.hbm_cold` is a second timed leg in the same
run that defeats cache residency at the same launch size (round-robin over enough disjoint buffer sets
to exceed twice the Infinity Cache): that one is a true HBM fraction.  `roofline.valu` keeps the
instruction-issue analysis (what the resident kernel is nearest to on chip).

Prints exactly one JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

ALGO_BYTES_PER_STEP = 38  # 16 board in + 1 action in + 16 board out + 4 reward out + 1 done out
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290
INFINITY_CACHE_BYTES = 256 << 20
# VALU issue ceiling of the step kernel's OWN instruction mix, measured: tools/issue_probe (per-opcode cycles per wave64
# instruction per SIMD at 1/2/4/8 waves per SIMD, on an MI355X) weighted with the kernel's opcode histogram by
# tools/valu_ceiling.py -> profiles/valu_ceiling.json.  A 2^20-board launch of the four-boards-per-lane tiling puts 4 waves on
# every SIMD.  The fallback is the figure of the committed table (2.85 cycles per instruction -> 862 G wave-instr/s).
VALU_PEAK_GINSTR_FALLBACK = 256 * 4 * 2.4 / 2.85
SEED = 0x2048
KERNEL_SOURCES = ['simulate_2048_amd/csrc/' + name for name in ('step_kernels.hip', 'board_ops.hpp', 'board_io.hpp', 'philox_strip.hpp',
                                                                  'tiling.hpp', 'step_planner.hpp')]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--boards', type=int, default=1 << 20, help='boards per GPU')
    ap.add_argument('--min-ms', type=float, default=60.0, help='repeat the K-step block until this much GPU time is covered')
    ap.add_argument('--max-blocks', type=int, default=400)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target duration of the CPU baseline sample')
    ap.add_argument('--no-cold', action='store_true', help='skip the HBM-cold leg')
    ap.add_argument('--only-cold', action='store_true', help='profiling aid: run ONLY the HBM-cold leg (so that rocprofv3 counters of '
                    'step_kernel<false, false> describe cold launches) and print its object')
    ap.add_argument('--cold-stores', choices=['auto', 'through', 'back'], default='auto', help='store-policy hint of the cold leg')
    ap.add_argument('--cold-tile', choices=['auto', 'small', 'large'], default='auto', help='tiling hint of the cold leg')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the BASELINE configs[1] / configs[4] legs')
    ap.add_argument('--extras', action='store_true', help='also emit legal mask + obs (not the headline)')
    ap.add_argument('--obs-dtype', choices=['f32', 'f16', 'bf16'], default='f32', help='with --extras: element type of obs')
    ap.add_argument('--single-process', action='store_true', help='ONE process drives all --gpus devices through ShardedEnv2048 (a worker '
                    'thread + stream per device) instead of one rank per GPU; no torch.distributed at all')
    ap.add_argument('--shards-on-device-0', action='store_true', help='with --single-process: put all --gpus shards on device 0 (REHEARSAL of '
                    'the fan-out on a box with fewer GPUs; not a scaling measurement)')
    ap.add_argument('--no-direct', action='store_true', help='A/B aid: ENV2048_FLAG_NO_DIRECT on the timed env2048_step_sequence calls (the '
                    'stream route: HIP launches instead of the library\'s own queues)')
    ap.add_argument('--single-chain', action='store_true', help='A/B aid: ENV2048_FLAG_SINGLE_CHAIN on the timed env2048_step_sequence calls '
                    '(never split the batch into two half-batch chains)')
    ap.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL, one GPU per rank); 'gloo' = REHEARSAL of the "
                    'multi-rank logic on a box with fewer GPUs than ranks (all ranks then share device 0; numbers are not a '
                    'scaling measurement and the JSON says so)')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(world: int, worker_cmd: list[str], timeout: float = 1500.0) -> int:
    """Start `world` worker processes (this process has made no GPU call and makes none), wait for all,
    relay rank 0's stdout.  Any failing rank => the others are terminated (exact PIDs) and the exit code is
    non-zero.  Nothing here replaces a running process image (no exec)."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '1')
        procs.append(subprocess.Popen(worker_cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + timeout
    rc = 0
    pending = set(range(world))
    out0 = ''
    while pending:
        for rank in sorted(pending):
            p = procs[rank]
            if rank == 0:
                try:
                    out0_part, _ = p.communicate(timeout=0.2)
                    out0 += out0_part or ''
                except subprocess.TimeoutExpired:
                    continue
            elif p.poll() is None:
                continue
            pending.discard(rank)
            if p.returncode != 0:
                rc = rc or p.returncode or 1
                print(f'bench.py: rank {rank} exited with code {p.returncode}', file=sys.stderr)
        if (rc or time.monotonic() > deadline) and pending:
            rc = rc or 124
            for rank in pending:
                procs[rank].terminate()
            for rank in pending:
                try:
                    procs[rank].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[rank].kill()
            break
        time.sleep(0.05)
    for line in out0.splitlines():  # exactly the JSON line on stdout; anything else a library printed there goes to stderr
        print(line, file=sys.stdout if line.startswith('{') else sys.stderr, flush=True)
    if rc == 0 and not any(line.startswith('{') for line in out0.splitlines()):
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
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
    single_elapsed = 0.0
    while threads > 1:
        oracle.step_batch_inplace(boards, actions, reward, done, SEED, 500_000 + single_steps, flags=1, threads=1)
        single_steps += 1
        single_elapsed = time.perf_counter() - t1
        if single_elapsed >= 2.0:
            break
    single = f'; 1 thread: {n * single_steps / single_elapsed:.2e} steps/s' if single_steps else ''
    # the reference-shaped per-board NumPy engine (oracle/numpy_shaped.py: same algorithmic structure as the reference's
    # gameboard.py, which cannot travel to this box) on one core, ~2 s of random play on one board
    from oracle import numpy_shaped

    board = oracle.unpack_values(boards_np[0]).reshape(4, 4)
    shaped_steps, t2 = 0, time.perf_counter()
    while True:
        word = oracle.step_word(SEED, 0, 900_000 + shaped_steps)
        board, _, finished = numpy_shaped.transition(board, word & 3, word)
        if finished:
            board = oracle.unpack_values(boards_np[(shaped_steps + 1) % n]).reshape(4, 4)
        shaped_steps += 1
        shaped_elapsed = time.perf_counter() - t2
        if shaped_elapsed >= 2.0:
            break
    return {
        'value': n * steps / elapsed,
        'unit': 'board-steps/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'{steps} steps x {n} boards (same mid-game boards, auto-reset, host uint8 actions) in {elapsed:.1f} s{single}; '
                  f'oracle/env2048_oracle.c, gcc -O2 -fopenmp; reference NumPy engine itself: ~1.0e4 steps/s/core (BASELINE.md)',
        'reference_shaped_numpy': {
            'value': shaped_steps / shaped_elapsed, 'unit': 'board-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{shaped_steps} single-board steps in {shaped_elapsed:.1f} s with oracle/numpy_shaped.py -- a per-board NumPy engine with '
                      f'the algorithmic structure of the reference\'s gameboard.py (rot90, per-row Python loop, argwhere spawn), checked against the '
                      f'reference-generated fixtures; the reference itself measured 9.5-10.2e3 steps/s on one core of the build container',
        },
    }


def kernel_source_sha() -> str:
    """Hash of what the step kernel is built from: the regions of step_kernels.hip and of the headers it is made of
    (KERNEL_SOURCES) between the [step-kernel-source-begin] / [step-kernel-source-end] markers (device code of step_kernel
    and its host launch policy), with comments and whitespace removed.  Edits to the other kernel families, or to comments,
    do not invalidate the committed counters."""
    import re

    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        text = (REPO / rel).read_text()
        regions = re.findall(r'\[step-kernel-source-begin\](.*?)\[step-kernel-source-end\]', text, flags=re.S)
        code = ''.join(regions) if regions else text
        code = re.sub(r'/\*.*?\*/', '', code, flags=re.S)       # block comments
        code = re.sub(r'//[^\n]*', '', code)                    # line comments (no string literal of these files contains //)
        h.update(''.join(code.split()).encode())
    return h.hexdigest()[:16]


def load_step_profile(boards: int, leg: str):
    """Counters of the step kernel from the committed rocprofv3 --pmc passes (profiles/step_pmc.json, written by
    tools/summarize_profiles.py): {'fabric_bytes_per_launch', 'valu_insts_per_launch', 'source', 'stale'} or None.
    'stale' = the kernel sources changed since the counters were collected."""
    path = REPO / 'profiles' / 'step_pmc.json'
    try:
        table = json.loads(path.read_text())
        entry = table['entries'].get(f'{leg}:{boards}')
        if entry is None:
            return None
        return dict(entry, stale=table.get('kernel_source_sha') != kernel_source_sha())
    except Exception:
        return None


def load_kernel_stats(leg: str):
    """rocprofv3 --kernel-trace --stats of the step kernel in the committed profile of this leg (profiles/step_pmc.json ->
    'kernel_stats', written by tools/summarize_profiles.py), with what that average means; None when there is none."""
    try:
        table = json.loads((REPO / 'profiles' / 'step_pmc.json').read_text())
        ks = dict(table['kernel_stats'][leg])
    except Exception:
        return None
    ks['stale'] = table.get('kernel_source_sha') != kernel_source_sha()
    if leg == 'direct':
        ks['note'] = ('direct route: two half-batch chains of env2048_step_resident_* launches on the library\'s own queues; the profiler '
                      'intercepts those queues like any other, its average is the duration of ONE half-batch kernel (two are in flight '
                      'in the timed run, so the step period is about one kernel duration)')
        return ks
    ks['note'] = ('one chain: the profiler isolates the whole-batch kernels; their average is the launch duration itself and is what '
                  'roofline.avg_launch_us of a one-chain run (K-step event time / K, which also holds the gaps between launches) has to '
                  'agree with' if leg != 'resident' else
                  'two half-batch chains: under the profiler every launch of the two enqueuing threads costs extra host time, the run '
                  'becomes launch-bound (its own step period: bench_event_us_per_step_in_the_profiled_run) and the half-batch kernels '
                  'mostly run alone -- their profiled average is the duration of a half-batch kernel with little overlap, NOT the step '
                  'period of the timed run (avg_launch_us, two kernels in flight).  The like-for-like check of event time against '
                  'rocprofv3 is the one-chain pair: step_pmc.json kernel_stats.resident_one_chain vs a --single-chain (or K < 32) run')
    return ks


def valu_ceiling(waves_per_simd: int = 4):
    """(G wave-instructions/s, source text, sustained rate of the real instruction stream or None) of the measured VALU ceiling
    for the step kernel's mix."""
    path = REPO / 'profiles' / 'valu_ceiling.json'
    try:
        table = json.loads(path.read_text())
        row = table['ceiling'][str(waves_per_simd)]
        real = table.get('real_instruction_stream')
        return row['g_wave_instr_per_s'], (f'profiles/valu_ceiling.json: {row["cycles_per_wave_instruction"]} cycles per wave64 instruction for this '
                                           f'kernel\'s opcode mix at {waves_per_simd} waves per SIMD (tools/issue_probe on MI355X: {table["probe"]}), '
                                           f'1024 SIMDs x 2.4 GHz'), real
    except Exception:
        return VALU_PEAK_GINSTR_FALLBACK, 'fallback constant of bench.py (profiles/valu_ceiling.json not readable)', None


def build_roofline(args, lib, n: int, flags: int, launch_s: float, obs_bytes: int, route=None) -> dict:
    """The roofline object of a run whose dominant kernel is one env2048_step launch over n boards taking launch_s:
    SURVEY 8(d)'s definition at the top level (algorithmic bytes / launch time against the HBM peak), the VALU-issue analysis
    of the cache-resident regime under `valu`."""
    bytes_per_launch = ALGO_BYTES_PER_STEP * n + (n * (1 + 16 * obs_bytes) if args.extras else 0)
    algo_gbs = bytes_per_launch / launch_s / 1e9
    working_set = n * (16 + 8 + 4 + 1) + (n * (1 + 16 * obs_bytes) if args.extras else 0)  # boards + 8-array action pool + outputs
    resident = working_set <= INFINITY_CACHE_BYTES
    # the library's own launch policy (env2048_step_sequence_plan), not a copy of its size rules: the timed region is one
    # env2048_step_sequence call, which may run the batch as two half-batch chains on two streams
    plan = lib.step_sequence_plan(n, args.steps, flags)
    direct = bool(route and route.get('direct'))
    if direct:  # the library's own queues: two half-batch chains of resident launches (csrc/direct_dispatch.hpp)
        plan = dict(plan, route='direct', chains=route['chains'], write_through=False,
                    boards_per_lane=lib.step_plan(n if route['chains'] == 1 else n - (n // 2) // 2048 * 2048, 0)['boards_per_lane'])
        if route.get('device_s_per_step') and not 0.5 * launch_s <= route['device_s_per_step'] <= 1.02 * launch_s:
            route = dict(route, device_s_per_step=None, device_time_rejected='the dispatch timestamps disagree with the event time around the call')
        if route.get('device_s_per_step'):
            event_s, launch_s = launch_s, route['device_s_per_step']
    else:
        plan = dict(plan, route='stream')
    leg = 'direct' if direct else 'resident' if plan['chains'] == 2 else 'resident_one_chain'  # counters / kernel stats of the SHAPE that is timed
    algo_gbs = bytes_per_launch / launch_s / 1e9
    prof = None if args.extras else load_step_profile(n, leg)
    tiling = plan['boards_per_lane']
    kernel = (f'env2048_step_resident_{"large" if tiling == 4 else "small"}' if direct else
              f'step_kernel<false, false, {tiling}>' if not args.extras else f'step_kernel<true, false, {tiling}>')
    roofline = {
        'bound': 'hbm',
        'achieved': algo_gbs,
        'peak': HBM_PEAK_GBS,
        'unit': 'GB/s',
        'frac': algo_gbs / HBM_PEAK_GBS,
        'regime': (f'L2-resident: between the first and the last launch of a sequence the {n * 16 / 2**20:.0f} MiB of boards never leave the eight '
                   f'4 MiB XCD L2s (no cache fence on the packets between; `traffic` counts what still crosses the fabric per step: the action '
                   f'row in, the reward / done rows out), so this fraction (algorithmic bytes / device-side step period / 8 TB/s, SURVEY 8(d)) '
                   f'is an EQUIVALENT, not bytes that crossed the HBM pins; the true HBM fraction of the step kernel is roofline.hbm_cold'
                   if direct else
                   f'cache-resident: the {working_set / 2**20:.0f} MiB working set of back-to-back steps lives in the 256 MiB Infinity Cache, so this '
                   f'fraction (algorithmic bytes / launch time / 8 TB/s, SURVEY 8(d)) is an EQUIVALENT, not bytes that crossed the HBM pins; the '
                   f'true HBM fraction of this kernel is roofline.hbm_cold' if resident else
                   'HBM: the working set exceeds the Infinity Cache'),
        'kernel': kernel,
        'launch_plan': plan,
        'launches_per_step': plan['chains'],
        'boards_per_kernel_launch': n if plan['chains'] == 1 else n - (n // 2) // 2048 * 2048,
        'avg_launch_us': launch_s * 1e6,
        'avg_launch_us_means': ('device-side period of one step over the whole batch: (end of the last packet - start of the first packet) / K from the '
                                'dispatch timestamps of the library\'s queues (env2048_direct_stats.last_gpu_ns; HIP events cannot be recorded on those '
                                'queues -- the event time of the region around the synchronous call is timing.event_us_per_step); the step is two '
                                'half-batch launches on two queues that overlap each other' if direct and route.get('device_s_per_step') else
                                'HIP-event time of the K-step region / K = the period of one step over the whole batch' +
                                ('' if plan['chains'] == 1 else '; the step is two half-batch launches on two streams that overlap each '
                                 'other, each lasting about one period (the rocprofv3 average of the half-batch kernel)')),
        'algorithmic_bytes_per_launch': bytes_per_launch,
        'traffic': None if prof is None else prof.get('fabric_bytes_per_launch'),  # (the table is scaled to one step over the whole batch)
        'traffic_kind': 'fabric bytes per step over the whole batch (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of the timed '
                        'shape): L2 <-> Infinity Cache/HBM requests -- Infinity-Cache hits are counted, so this is HBM traffic only for the cold leg',
        'traffic_source': None if prof is None else dict(source=prof.get('source'), leg=leg, stale=prof['stale']),
        'rocprof_kernel_stats': load_kernel_stats(leg),
    }
    if direct:
        roofline['route'] = {'direct': True, 'reruns': route.get('reruns'), 'state': route.get('state'),
                             'event_us_per_step_around_the_call': event_s * 1e6 if route.get('device_s_per_step') else None}
    if resident and prof is not None and prof.get('valu_insts_per_launch'):
        valu_per_step = prof['valu_insts_per_launch']  # (scaled to one step over the whole batch by tools/summarize_profiles.py)
        ginstr = valu_per_step / launch_s / 1e9
        waves_per_simd = max(1, min(8, round(n / (64 * tiling) / 1024)))  # both half-batch chains are resident together
        peak, peak_source, real_stream = valu_ceiling(4 if waves_per_simd >= 4 else 2 if waves_per_simd >= 2 else 1)
        per_lane = valu_per_step * 64 / n  # wave64 instructions x 64 lanes / boards
        roofline['valu'] = {
            'achieved': ginstr,
            'peak': peak,
            'peak_source': peak_source,
            'unit': 'G wave-instr/s',
            'frac': ginstr / peak,
            'why': f'of the on-chip limits of the cache-resident regime VALU issue is the nearest: {per_lane:.0f} VALU instructions per board-step '
                   f'(SQ_INSTS_VALU {valu_per_step:.0f} wave64 instructions per step = {valu_per_step / n:.2f} per board, '
                   f'x 64 lanes / board), against the MEASURED issue ceiling of this opcode mix at {waves_per_simd} waves per SIMD.  '
                   f'The launch is 15-20 % above its memory-only floor (tools/microbench, DESIGN 6.1)',
            'vs_real_instruction_stream': None if not real_stream else {
                'sustained_g_wave_instr_per_s': real_stream['saturated_g_wave_instr_per_s'],
                'frac': ginstr / real_stream['saturated_g_wave_instr_per_s'], 'source': real_stream['source'],
                'note': 'what the SAME instruction stream sustains with the boards in registers (env2048_rollout, any occupancy from 8 waves '
                        'per SIMD): the practical VALU ceiling of this code; `peak` above is the stricter single-opcode-probe bound'},
        }
    return roofline


def workload_text(n: int) -> str:
    which = {1 << 16: 'BASELINE configs[1]', 1 << 20: 'BASELINE configs[2]'}.get(n, 'BASELINE configs[2] workload at a non-default size')
    count = f'{n:,}'.replace(',', ' ')
    return (f'{which}: {count} parallel 4x4 boards per GPU, uint8 log2-packed, uniform random host actions, Philox spawn, '
            f'terminal check, auto-reset; boards mid-game (64 warm-up steps)')


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
class RankComm:
    """The only collective traffic of the benchmark: the barrier that brackets a timed block, the max over ranks of the block
    times and the gather of the per-rank medians (never on the hot path).  `backend` None = a single rank without a process
    group; 'nccl' = RCCL with the reductions on `device`; 'gloo' = CPU tensors (rehearsals, CPU tests)."""

    def __init__(self, backend, device=None):
        import torch

        self.torch = torch
        self.dist = None
        self.world = 1
        self.reduce_device = 'cpu'
        if backend is not None:
            import torch.distributed as dist

            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=device)
                self.reduce_device = device
            else:
                dist.init_process_group(backend)
            self.dist = dist
            self.world = dist.get_world_size()

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def reduce_max(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.reduce_device)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().tolist()

    def gather(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.reduce_device)
        if self.dist is None:
            return [t.tolist()]
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.cpu().tolist() for o in out]

    def teardown(self) -> None:
        if self.dist is not None:
            try:
                self.dist.barrier()
            finally:
                self.dist.destroy_process_group()


def reduce_timings(comm: RankComm, walls, events, K: int) -> dict:
    """What rank 0 reports from every rank's per-block times: per block the SLOWEST rank (max over ranks), then the median
    block; plus every rank's own medians."""
    walls_max = comm.reduce_max(walls)
    events_max = comm.reduce_max(events)
    per_rank = comm.gather([statistics.median(events) / K * 1e6, statistics.median(walls) / K * 1e6])
    return {'walls_max': walls_max, 'events_max': events_max, 'per_rank': per_rank, 'wall': statistics.median(walls_max),
            'launch_s': statistics.median(events_max) / K}


def assemble_result(args, world: int, n: int, timings: dict, blocks: int, enqueue_us: float, roofline: dict, *, rehearsal: bool = False,
                    other=None, cpu=None, route=None) -> dict:
    """The one JSON object rank 0 prints (the driver's contract): `value` = boards x K x ranks / the median block's wall time
    (per block: the slowest rank)."""
    K, W = args.steps, args.warmup
    wall, launch_s, walls_max, events_max = timings['wall'], timings['launch_s'], timings['walls_max'], timings['events_max']
    out = {
        'metric': 'env_board_steps_per_sec',
        'value': float(n) * K * world / wall,
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
            'workload': workload_text(n),
            'boards_per_gpu': n,
            'global_boards': n * world,
            'launch_mode': ('env2048_step_sequence, direct route: one call writes the K kernel-dispatch packets of each of two half-batch chains into '
                            'the library\'s own user-mode queues (no HIP launch, boards resident in the XCD L2s between the first and the last '
                            'packet) and returns when they have completed' if route and route.get('direct') else
                            'env2048_step_sequence: the K steps enqueued from C by one call (two half-batch chains of K launches each when the '
                            'library plans it so: roofline.launch_plan)'),
            'sharding': 'contiguous env-index slices, one process per GPU, no collective on the hot path',
            'outputs': 'board,reward,done' + (f',legal,obs({args.obs_dtype})' if args.extras else ''),
        },
        'timing': {
            'blocks': blocks,
            'block_steps': K,
            'wall_us_per_step': {'median': wall / K * 1e6, 'min': min(walls_max) / K * 1e6, 'max': max(walls_max) / K * 1e6},
            'event_us_per_step': {'median': launch_s * 1e6, 'min': min(events_max) / K * 1e6, 'max': max(events_max) / K * 1e6},
            'per_rank_us_per_step': [{'rank': r, 'event': v[0], 'wall': v[1]} for r, v in enumerate(timings['per_rank'])],
            'host_enqueue_us_per_launch': enqueue_us,  # rank 0; meaningful while the launch queue has room (small K)
            'note': 'value = boards x K x ranks / median over blocks of (max over ranks of the block wall time); wall-clock blocks '
                    'and HIP-event blocks alternate (an event-timed block is not wall-timed: recording events is itself host work)',
        },
        'roofline': roofline,
    }
    if rehearsal:
        out['config']['rehearsal'] = (f'{world} gloo ranks SHARING one GPU: exercises the multi-rank code path only; '
                                      f'not a scaling measurement')
    if other is not None:
        out['other_configs'] = other
    if cpu is not None:
        out['cpu_baseline'] = cpu
    elif world > 1:
        out['cpu_baseline'] = None
        out['cpu_baseline_note'] = 'timed on rank 0 of the N = 1 run only (the contract); run bench.py --gpus 1 for it'
    else:
        out['cpu_baseline'] = None
        out['cpu_baseline_note'] = 'skipped by --no-cpu-baseline'
    return out


def run_rank(args) -> None:
    # Only the JSON line may reach stdout: libraries print there too (RCCL's version banner on its first collective,
    # gloo's connection notes), so file descriptor 1 is pointed at stderr for the whole run and the result line is
    # written to the saved original at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        os.write(result_fd, (line + '\n').encode())

    import torch

    from simulate_2048_amd import capi
    from simulate_2048_amd.batch import BatchedEnv2048

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a HIP device: the product path has no CPU fallback')
    rehearsal = world > 1 and args.dist_backend != 'nccl'
    device_index = 0 if rehearsal else local_rank
    if device_index >= torch.cuda.device_count():
        sys.exit(f'rank {rank}: local rank {local_rank} has no GPU (visible devices: {torch.cuda.device_count()}); '
                 f'use --dist-backend gloo to REHEARSE the multi-rank logic on fewer GPUs')
    torch.cuda.set_device(device_index)
    device = torch.device('cuda', device_index)
    # ENV2048_BENCH_FORCE_DIST=1: initialise the process group even for one rank (started with RANK=0 WORLD_SIZE=1
    # MASTER_ADDR/PORT set) -- a one-GPU rehearsal of the RCCL barrier / all_reduce / all_gather calls of the N > 1 path
    comm = RankComm(args.dist_backend if (world > 1 or os.environ.get('ENV2048_BENCH_FORCE_DIST') == '1') else None, device)
    barrier, teardown = comm.barrier, comm.teardown

    def body():
        n = args.boards
        K, W = args.steps, args.warmup
        obs_bytes = 4 if args.obs_dtype == 'f32' else 2
        obs_dtype = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[args.obs_dtype]
        env_base = rank * n
        env = BatchedEnv2048(n, SEED, device=device, env_base=env_base, auto_reset=True, with_legal=args.extras,
                             with_obs=args.extras, obs_dtype=obs_dtype)
        env.reset()
        for _ in range(64):  # bring the boards to mid-game
            env.step(None)
        torch.cuda.synchronize()
        midgame = env.boards.clone()
        boards_midgame = midgame.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

        # host-provided actions: a small pool of uint8 arrays, cycled
        gen = torch.Generator(device='cpu').manual_seed(1 + rank)
        pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).to(device) for _ in range(8)]

        lib = capi.library()
        stream = torch.cuda.current_stream(device)
        h = stream.cuda_stream
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        flags = env.flags | (capi.FLAG_SINGLE_CHAIN if args.single_chain else 0) | (capi.FLAG_NO_DIRECT if args.no_direct else 0)

        tables = {}

        def action_table(count: int, first: int):
            """Host array of `count` device pointers cycling through the action pool (built once per (count, phase): the timed
            region then holds the C call and nothing else)."""
            key = (count, first & 7)
            if key not in tables:
                tables[key] = (ctypes.c_void_p * count)(*[pool[(first + t) & 7].data_ptr() for t in range(count)])
            return tables[key]

        step_index = env.step_index
        fixed = (p(env.boards), p(env.reward), p(env.done), p(env.legal), p(env.obs))
        step_sequence = lib._dll.env2048_step_sequence  # the raw entry point: argument conversion is all the Python left per block

        def run_steps(count: int) -> None:
            """`count` dependent step launches, enqueued from C by one call."""
            nonlocal step_index
            if count <= 0:
                return
            rc = step_sequence(device_index, h, n, env_base, fixed[0], action_table(count, step_index), fixed[1], fixed[2], fixed[3], fixed[4],
                               0, SEED, step_index, count, flags)
            if rc != 0:
                raise capi.Env2048Error(rc, lib.last_error())
            step_index += count

        for phase in range(8):
            action_table(K, phase)
        enqueue_times = []
        route_calls, device_times = [0], []  # (filled in below: querying the direct route starts it)

        def timed_block(count: int, with_events: bool = True):
            """One timed region: barrier + synchronize | K steps | synchronize.  Blocks alternate between two instruments: the
            host clock alone (what `value` is computed from: nothing but the K steps sits between the two synchronisations) and
            HIP events on the launch stream (the GPU-side duration, for the roofline); recording the events costs a few
            microseconds of host time and two marker packets per block, which a 20-step block would otherwise carry as load."""
            ev0 = ev1 = None
            if with_events:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if with_events:
                ev0.record(stream)
            run_steps(count)
            enqueued = time.perf_counter() - t0
            if with_events:
                ev1.record(stream)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            enqueue_times.append(enqueued / max(count, 1))
            if with_events:  # (outside the timed region) the route the call took and, on the direct route, its duration on the device
                info = lib.direct_stats(device_index)
                route_calls.append(info['calls'])
                device_times.append(info['last_gpu_ns'] * 1e-9 if len(route_calls) > 1 and route_calls[-1] > route_calls[-2] else 0.0)
            return wall, (ev0.elapsed_time(ev1) * 1e-3 if with_events else None)

        if args.only_cold:
            cold = cold_leg(args, torch, lib, capi, device, device_index, h, stream, n, env_base, world, midgame, pool, flags, 8e-6)
            if rank == 0:
                emit(json.dumps({'hbm_cold': cold}))
            return
        stream_side = None
        if not args.no_other_configs and world == 1 and not args.extras and not args.no_direct:
            stream_side = stream_route_legs(torch, lib, device, device_index, h)  # (before anything can start the direct route)
        route_calls[0] = lib.direct_stats(device_index)['calls']
        run_steps(W)
        torch.cuda.synchronize()
        # how many blocks: cover --min-ms of GPU time (the same count on every rank: calibrate on the max over ranks)
        calib = comm.reduce_max([timed_block(K)[1]])[0]
        blocks = int(min(args.max_blocks, max(3, math.ceil(args.min_ms * 1e-3 / max(calib, 1e-7)))))
        walls, events = [], []
        for _ in range(blocks):
            walls.append(timed_block(K, with_events=False)[0])
            events.append(timed_block(K, with_events=True)[1])
        barrier()
        timings = reduce_timings(comm, walls, events, K)
        launch_s = timings['launch_s']
        # the route the timed calls took (env2048_direct_stats): on the direct route the launch duration is the DEVICE-side one
        info = lib.direct_stats(device_index)
        direct_blocks = [d for d in device_times[-blocks:] if d > 0]
        route = {'direct': len(direct_blocks) == blocks and info['calls'] > route_calls[0], 'chains': info['last_chains'], 'reruns': info['reruns'],
                 'state': info['reason'], 'device_s_per_step': (statistics.median(direct_blocks) / K) if direct_blocks else None}

        # ---- HBM-cold leg: same launch size, round-robin over disjoint buffer sets that exceed the Infinity Cache -------
        cold = None
        if not args.no_cold and not args.extras:
            cold = cold_leg(args, torch, lib, capi, device, device_index, h, stream, n, env_base, world, midgame, pool, flags, launch_s)

        # ---- BASELINE configs[1] and configs[4], single GPU only (reported next to the headline, not as `value`) ---------
        other = None
        if not args.no_other_configs and world == 1 and not args.extras:
            other = other_configs(torch, lib, device, device_index, h, stream_side=stream_side)

        if rank == 0:
            roofline = build_roofline(args, lib, n, flags, launch_s, obs_bytes, route=route)
            if cold is not None:
                roofline['hbm_cold'] = cold
            cpu = cpu_baseline(boards_midgame, args.cpu_seconds) if boards_midgame is not None else None
            emit(json.dumps(assemble_result(args, world, n, timings, blocks, statistics.median(enqueue_times) * 1e6, roofline,
                                            rehearsal=rehearsal, other=other, cpu=cpu, route=route)))

    try:
        body()
    finally:
        teardown()  # also on the --only-cold path and when a leg raises


def cold_leg(args, torch, lib, capi, device, device_index, h, stream, n, env_base, world, midgame, pool, flags, launch_s) -> dict:
    """The step kernel at the same launch size with cache residency defeated: `sets` disjoint (boards, actions, reward,
    done) buffer sets whose total exceeds twice the 256 MiB Infinity Cache, stepped round-robin, so every launch reads
    and writes lines that were last touched `sets` launches ago."""
    set_bytes = n * (16 + 1 + 4 + 1)
    sets = max(2, math.ceil(2.5 * INFINITY_CACHE_BYTES / set_bytes))
    cb = midgame.unsqueeze(0).repeat(sets, 1, 1).contiguous()
    ca = torch.stack([pool[s & 7] for s in range(sets)]).contiguous()
    cr = torch.zeros((sets, n), dtype=torch.int32, device=device)
    cd = torch.zeros((sets, n), dtype=torch.uint8, device=device)
    records = (capi.StepArgs * sets)()
    cold_step = 1 << 32
    hint = {'auto': 0, 'through': capi.FLAG_STORES_WRITE_THROUGH, 'back': capi.FLAG_STORES_WRITE_BACK}[args.cold_stores]
    hint |= {'auto': 0, 'small': capi.FLAG_TILE_SMALL, 'large': capi.FLAG_TILE_LARGE}[args.cold_tile]
    def cold_pass() -> None:
        nonlocal cold_step
        for s in range(sets):
            records[s] = capi.StepArgs(n, (world + s) * n + env_base, cb[s].data_ptr(), ca[s].data_ptr(), cr[s].data_ptr(),
                                       cd[s].data_ptr(), None, None, SEED, cold_step, flags | hint, 0)
        lib.step_many(device_index, h, records)
        cold_step += 1

    for _ in range(3):
        cold_pass()
    torch.cuda.synchronize()
    plan = lib.step_many_plan(records)
    passes = max(4, min(64, math.ceil(25e-3 / max(launch_s * sets, 1e-7))))

    def time_passes(one_pass) -> float:
        times = []
        for _ in range(3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            for _ in range(passes):
                one_pass()
            ev1.record(stream)
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e-3 / (passes * sets))
        return statistics.median(times)

    cold_s = time_passes(cold_pass)
    # the same records one env2048_step launch each (what round 2 measured, and what a DEPENDENT chain of cold launches pays):
    # every launch is load-everything -> compute -> store-everything -> kernel boundary
    single = (capi.StepArgs * 1)()

    def separate_pass() -> None:
        nonlocal cold_step
        for s in range(sets):
            single[0] = capi.StepArgs(n, (world + s) * n + env_base, cb[s].data_ptr(), ca[s].data_ptr(), cr[s].data_ptr(), cd[s].data_ptr(),
                                      None, None, SEED, cold_step, flags | hint, 0)
            lib.step_many(device_index, h, single)
        cold_step += 1

    separate_pass()
    separate_s = time_passes(separate_pass)
    copy_gbs = measured_copy_ceiling(torch, device)
    prof = load_step_profile(n, 'cold')
    gbs = lambda t: ALGO_BYTES_PER_STEP * n / t / 1e9
    return {
        'bound': 'hbm',
        'achieved': gbs(cold_s),
        'peak': HBM_PEAK_GBS,
        'unit': 'GB/s',
        'frac': gbs(cold_s) / HBM_PEAK_GBS,
        'measured_copy_ceiling': None if copy_gbs is None else {
            'GB/s': copy_gbs['GB/s'], 'frac_of_it': gbs(cold_s) / copy_gbs['GB/s'], 'shape': copy_gbs['shape'], 'sweep_GB/s': copy_gbs['sweep_GB/s'],
            'what': 'SURVEY 8(d): the device-copy ceiling measured on THIS box in this run beside the 8 TB/s spec peak -- a 1 GiB streaming copy '
                    '(tools/copy_probe.hip; read + written bytes / HIP-event time, median of 5 x 10 copies, 2 GiB touched per copy > 8 x Infinity '
                    'Cache; best of a small sweep of launch shapes)'},
        'avg_launch_us': cold_s * 1e6,
        'avg_launch_us_means': f'time per {n}-board record; the {sets} records of a call share {plan["launches"]} kernel launch(es)',
        'kernel': f'step_many_kernel<false, false, {plan["boards_per_lane"][0]}>' if plan['launches'] < sets else f'step_kernel<false, false, {plan["boards_per_lane"][0]}>',
        'launch_plan': {'records_per_call': sets, 'kernel_launches_per_call': plan['launches'], 'boards_per_lane': plan['boards_per_lane'][0],
                        'write_through': plan['write_through'][0],
                        'policy': 'env2048_step_many: consecutive records that touch disjoint memory share one launch (no caller hint)'},
        'buffer_sets': sets,
        'footprint_bytes': sets * set_bytes,
        'records_timed': passes * sets,
        'stores': args.cold_stores,
        'tile': args.cold_tile,
        'separate_launches': {'us_per_launch': separate_s * 1e6, 'achieved': gbs(separate_s), 'frac': gbs(separate_s) / HBM_PEAK_GBS,
                              'note': 'the same records as one kernel launch each (round 2\'s figure; what a dependent chain of cold '
                                      'launches pays: a pipeline fill and drain per launch)'},
        'traffic': None if prof is None else prof.get('fabric_bytes_per_launch'),
        'traffic_source': None if prof is None else dict(source=prof.get('source'), stale=prof['stale']),
        'how': f'{sets} disjoint sets of (boards, actions, reward, done) = {sets * set_bytes / 2**20:.0f} MiB > 2 x Infinity Cache, '
               f'stepped round-robin by env2048_step_many, so every record reads and writes lines last touched {sets} records ago',
    }


def measured_copy_ceiling(torch, device):
    """GB/s (read + written) of a plain 1 GiB device-to-device streaming copy (tools/copy_probe.hip: 16-byte vectors, four
    loads in flight per lane): what this box's HBM sustains for a pure stream, the practical ceiling an HBM-bound kernel is
    priced against beside the spec peak.  None when the probe library cannot be built or loaded."""
    import ctypes

    try:
        import __graft_entry__

        probe = ctypes.CDLL(str(__graft_entry__.build_copy_probe()))
    except Exception as exc:  # noqa: BLE001 (a missing measuring stick must not cost the bench line)
        print(f'bench: copy probe unavailable ({exc})', file=sys.stderr)
        return None
    probe.copy_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
    size = 1 << 30
    src = torch.zeros((size,), dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    stream = torch.cuda.current_stream(device)
    best = None
    shapes = [(w, nt) for nt in (0, 1) for w in (0, 4096, 16384)]  # (0 workgroups: one 16 KiB tile per workgroup)
    if os.environ.get('ENV2048_COPY_PROBE_SWEEP'):
        shapes = [(w, nt) for nt in (0, 1) for w in (0, 1024, 2048, 4096, 8192, 16384, 65536)]
    sweep = {}
    for workgroups, nt in shapes:
        assert probe.copy_probe(stream.cuda_stream, src.data_ptr(), dst.data_ptr(), size, workgroups, nt) == 0
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            for _ in range(10):
                probe.copy_probe(stream.cuda_stream, src.data_ptr(), dst.data_ptr(), size, workgroups, nt)
            ev1.record(stream)
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e-3 / 10)
        gbs = 2 * size / statistics.median(times) / 1e9
        sweep[f'{"tiles" if workgroups == 0 else workgroups}{" nt" if nt else ""}'] = round(gbs, 1)
        if best is None or gbs > best[0]:
            best = (gbs, workgroups, nt)
    del src, dst
    return {'GB/s': best[0], 'shape': f'{"one 16 KiB tile per workgroup" if best[1] == 0 else str(best[1]) + " workgroups, grid-stride"}, '
                                      f'{"non-temporal" if best[2] else "temporal"} accesses', 'sweep_GB/s': sweep}


def other_configs(torch, lib, device, device_index, h, stream_side=None) -> dict:
    """BASELINE configs[1] (65 536 boards) and configs[4] (2^20 boards x 512-step rollout, Philox spawn, auto-reset,
    on-device random actions) on this GPU -- measured both as 512 step launches and as one fused env2048_rollout,
    with the cross-check SURVEY 8(d) config 5 asks for (identical boards, sum of rewards and episode counts)."""
    from simulate_2048_amd import capi
    from simulate_2048_amd.batch import BatchedEnv2048

    out = {}
    stream_side = stream_side or {}
    # configs[0]: one board, 1000 random-action steps through the TwentyFortyEight env API (reset on done) -- the mirror runs
    # every step as one kernel launch on pinned host operands + one polled completion word
    import numpy as np

    from simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight import TwentyFortyEight

    single = TwentyFortyEight()
    single.reset(seed=0)
    picks = np.random.default_rng(0).integers(0, 4, 1000)
    for a in picks[:50]:
        single.step(int(a))
    t0 = time.perf_counter()
    for a in picks:
        _, _, finished = single.step(int(a))
        if finished:
            single.reset()
    dt = time.perf_counter() - t0
    out['configs[0]'] = {
        'workload': 'single 4x4 board, 1000 random-action steps via the TwentyFortyEight env API (mirror on the GPU: one launch on pinned '
                    'host operands + one polled completion word per step)',
        'value': 1000 / dt, 'unit': 'board-steps/s', 'us_per_step': dt * 1e3,
        'reference': 'the reference NumPy env: 9.5-10.2e3 steps/s on one core of the build container (BASELINE.md section 2)',
    }
    # configs[1]
    n1 = 1 << 16
    env = BatchedEnv2048(n1, SEED, device=device, auto_reset=True)
    env.reset()
    for _ in range(64):
        env.step(None)
    gen = torch.Generator(device='cpu').manual_seed(7)
    pool = [torch.randint(0, 4, (n1,), generator=gen, dtype=torch.uint8).to(device) for _ in range(8)]
    count = 2000
    table = (ctypes.c_void_p * count)(*[pool[t & 7].data_ptr() for t in range(count)])
    stream = torch.cuda.current_stream(device)
    times = []
    calls_before = lib.direct_stats(device_index)['calls']
    for rep in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        lib.step_sequence(device_index, h, n1, 0, env.boards.data_ptr(), table, env.reward.data_ptr(), env.done.data_ptr(), None, None,
                          0, SEED, 64 + rep * count, count, env.flags)
        ev1.record(stream)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
    wall, ev = sorted(times)[len(times) // 2]
    route1 = 'direct' if lib.direct_stats(device_index)['calls'] > calls_before else 'stream'
    # the same per-step launches as a REPLAYABLE hipGraph: the step index lives in device memory (step_counter), so every
    # replay continues the random stream; the host then costs one graph launch per 200 steps instead of 200 kernel launches
    genv = BatchedEnv2048(n1, SEED, device=device, auto_reset=True, device_step_counter=True)
    genv.boards.copy_(env.boards)
    genv.step_index = 10_000
    side = torch.cuda.Stream(device)
    graph = torch.cuda.CUDAGraph()
    captured = 200
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for t in range(captured):
                genv.step(pool[t & 7])
            genv.commit_steps()  # last node of the graph: device counter += captured
    torch.cuda.synchronize()
    graph.replay()
    graph_times = []
    for rep in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(10):
            graph.replay()
        ev1.record(stream)
        torch.cuda.synchronize()
        graph_times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
    gwall, gev = sorted(graph_times)[len(graph_times) // 2]
    replayed = genv.step_index - 10_000
    # the same workload with the moves STAGED ([200, N] uint8 rows): BatchedEnv2048.step_many picks env2048_replay at this size --
    # one launch per 200 moves, boards in registers, 6 B per board-step, no per-move launch at all
    renv = BatchedEnv2048(n1, SEED, device=device, auto_reset=True)
    renv.boards.copy_(env.boards)
    renv.step_index = 20_000
    staged = torch.stack([pool[t & 7] for t in range(captured)]).contiguous()
    rr = torch.zeros((captured, n1), dtype=torch.int32, device=device)
    dd = torch.zeros((captured, n1), dtype=torch.uint8, device=device)
    renv.step_many(staged, out_reward=rr, out_done=dd)
    replay_times = []
    for rep in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(10):
            renv.step_many(staged, out_reward=rr, out_done=dd)
        ev1.record(stream)
        torch.cuda.synchronize()
        replay_times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
    rwall, rev = sorted(replay_times)[len(replay_times) // 2]
    out['configs[1]'] = {
        'workload': '65 536 parallel boards, uint8 log2-packed, uniform random host actions, auto-reset, 1 GPU',
        'value': n1 * count / wall, 'unit': 'board-steps/s', 'us_per_step_wall': wall / count * 1e6, 'us_per_step_event': ev / count * 1e6,
        'route': route1 + (': one env2048_step_sequence call of 2000 steps on the library\'s own queue (one chain of resident launches, no cache '
                           'fence between them)' if route1 == 'direct' else ''),
        'as_stream_launches': stream_side.get('configs[1]'),
        'bound': 'launch rate: 1 MiB of boards is cache-resident and a launch (one-board-per-lane tiling, 128 workgroups) is shorter than '
                 'the command-processor cost of a dependent launch (and, on the stream route, than the host cost of enqueuing it: '
                 'timing.host_enqueue_us_per_launch)',
        'as_replayed_hipgraph': {'value': n1 * captured * 10 / gwall, 'unit': 'board-steps/s', 'us_per_step_wall': gwall / (captured * 10) * 1e6,
                                 'us_per_step_event': gev / (captured * 10) * 1e6, 'steps_replayed_check': replayed == captured * 41,
                                 'note': f'{captured} step launches + one counter-advance node captured once (device-side step counter), graph replayed: '
                                         f'no per-step host cost'},
        'as_staged_moves_one_replay_launch': {'value': n1 * captured * 10 / rwall, 'unit': 'board-steps/s',
                                              'us_per_step_wall': rwall / (captured * 10) * 1e6, 'us_per_step_event': rev / (captured * 10) * 1e6,
                                              'speedup_vs_launches': (wall / count) / (rwall / (captured * 10)),
                                              'note': f'BatchedEnv2048.step_many(actions[{captured}, N]) -> ONE env2048_replay launch per {captured} moves '
                                                      f'(reward / done rows written per move); the small-batch answer when the moves are known up front'},
    }
    del env, pool, genv, graph, renv, staged, rr, dd

    # configs[4]
    n4, T = 1 << 20, 512
    a = BatchedEnv2048(n4, SEED, device=device, auto_reset=True)
    a.reset()
    start = a.boards.clone()
    flags = capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS
    reward_rows = torch.zeros((T, n4), dtype=torch.int32, device=device)
    done_rows = torch.zeros((T, n4), dtype=torch.uint8, device=device)
    by_route = {}
    calls_before = lib.direct_stats(device_index)['calls']
    for route_name, extra in (('default', 0),):
        step_times = []
        for rep in range(3):  # (the first repetition also pays for the library's one-time second-chain stream / thread, if this is
            a.boards.copy_(start)  # the first long sequence of the process; the median of three does not)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record(stream)
            lib.step_sequence(device_index, h, n4, 0, a.boards.data_ptr(), None, reward_rows.data_ptr(), done_rows.data_ptr(), None, None,
                              n4, SEED, 0, T, flags | extra)
            ev1.record(stream)
            torch.cuda.synchronize()
            step_times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
        by_route[route_name] = sorted(step_times)[1]
    wall_steps, ev_steps = by_route['default']
    route4 = 'direct' if lib.direct_stats(device_index)['calls'] > calls_before else 'stream'
    sum_steps = int(reward_rows.sum(dtype=torch.int64))
    eps_steps = int(done_rows.sum(dtype=torch.int64))
    invalid = float(((reward_rows == 0) & (done_rows == 0)).float().mean())  # upper bound on the invalid-move share (valid moves can score 0)
    del reward_rows, done_rows
    b = BatchedEnv2048(n4, SEED, device=device, auto_reset=True)
    b.boards.copy_(start)
    roll_times = []
    stats = None
    for rep in range(3):
        b.boards.copy_(start)
        b.step_index = 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        stats = b.rollout(T)
        ev1.record(stream)
        torch.cuda.synchronize()
        roll_times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
    wall_roll, ev_roll = sorted(roll_times)[1]
    same_boards = bool(torch.equal(a.boards, b.boards))
    # the same 512 moves with PRE-STAGED host actions in one launch (env2048_replay): uniform random action rows
    gen = torch.Generator(device='cpu').manual_seed(11)
    rows = torch.randint(0, 4, (T, n4), generator=gen, dtype=torch.uint8).to(device)
    rr = torch.zeros((T, n4), dtype=torch.int32, device=device)
    dd = torch.zeros((T, n4), dtype=torch.uint8, device=device)
    replay_times = []
    for rep in range(3):
        b.boards.copy_(start)
        b.step_index = 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record(stream)
        b.replay(rows, out_reward=rr, out_done=dd)
        ev1.record(stream)
        torch.cuda.synchronize()
        replay_times.append(ev0.elapsed_time(ev1) * 1e-3)
    ev_replay = sorted(replay_times)[1]
    del rows, rr, dd
    sum_roll = int(stats['return_sum'].sum())
    eps_roll = int(stats['episodes'].sum(dtype=torch.int64))
    valid_roll = int(stats['valid_moves'].sum(dtype=torch.int64))
    out['configs[4]'] = {
        'workload': '1 048 576 boards x 512-step rollout, Philox tile spawn, on-device uniform random actions, auto-reset on terminal',
        'as_512_step_launches': {'value': n4 * T / wall_steps, 'unit': 'board-steps/s', 'ms_total_wall': wall_steps * 1e3,
                                 'us_per_step_event': ev_steps / T * 1e6, 'records': 'reward and done rows [512, 2^20] written in HBM',
                                 'route': route4 + (' (the library\'s own queues: 2 x 512 resident launches)' if route4 == 'direct' else ''),
                                 'on_the_stream_route': stream_side.get('configs[4]')},
        'as_one_rollout_kernel': {'value': n4 * T / wall_roll, 'unit': 'board-steps/s', 'ms_total_wall': wall_roll * 1e3,
                                  'ms_total_event': ev_roll * 1e3, 'bound': 'VALU issue (boards stay in registers: 44 B per env per 512 steps)'},
        'as_one_replay_kernel_host_actions': {'value': n4 * T / ev_replay, 'unit': 'board-steps/s', 'ms_total_event': ev_replay * 1e3,
                                              'note': 'env2048_replay: staged uint8 action rows in, reward + done rows [512, 2^20] out, '
                                                      '6 B per board-step (different actions than the two legs above)'},
        'cross_check': {'boards_identical': same_boards, 'reward_sum': [sum_steps, sum_roll], 'episodes': [eps_steps, eps_roll],
                        'ok': same_boards and sum_steps == sum_roll and eps_steps == eps_roll},
        'statistics': {'mean_episode_length': n4 * T / max(eps_roll, 1), 'invalid_move_fraction': 1.0 - valid_roll / (n4 * T),
                       'zero_reward_nonterminal_fraction': invalid},
    }
    del a, b
    out['configs[2]_long_sequence'] = long_sequence_leg(torch, lib, device, device_index, h, stream_side=stream_side)
    out['configs[2]_with_legal_obs'] = legal_obs_leg(torch, lib, device, device_index, h, stream_side=stream_side)
    return out


def _timed_sequence(torch, lib, stream, device_index, h, env, table, K, flags, step0, legal=None, obs=None, reps=5, calls=1):
    """Median (wall s, event s) PER CALL of `calls` consecutive env2048_step_sequence calls of K steps on `env`, timed as one region
    (after one untimed region of the same shape)."""
    times = []
    for rep in range(reps + 1):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for c in range(calls):
            lib.step_sequence(device_index, h, env.num_envs, env.env_base, env.boards.data_ptr(), table, env.reward.data_ptr(),
                              env.done.data_ptr(), legal, obs, 0, SEED, step0 + (rep * calls + c) * K, K, flags)
        ev1.record(stream)
        torch.cuda.synchronize()
        if rep:
            times.append(((time.perf_counter() - t0) / calls, ev0.elapsed_time(ev1) * 1e-3 / calls))
    return sorted(times)[len(times) // 2]


def _midgame_env(torch, device, n, **kw):
    from simulate_2048_amd.batch import BatchedEnv2048

    env = BatchedEnv2048(n, SEED, device=device, auto_reset=True, **kw)
    env.reset()
    for _ in range(64):
        env.step(None)
    gen = torch.Generator(device='cpu').manual_seed(1)
    pool = [torch.randint(0, 4, (n,), generator=gen, dtype=torch.uint8).to(device) for _ in range(8)]
    return env, pool


def long_sequence_leg(torch, lib, device, device_index, h, n: int = 1 << 20, K: int = 200, stream_side=None, stream_only: bool = False) -> dict:
    """The headline workload (BASELINE configs[2]) as ONE env2048_step_sequence call of K = 200 steps on every route the library
    has: the default (the direct route where it is available), the stream route as two half-batch chains (ENV2048_FLAG_NO_DIRECT)
    and as one chain (+ ENV2048_FLAG_SINGLE_CHAIN); and what the stream route's helper thread cost over a K = 2000 call."""
    from simulate_2048_amd import capi

    env, pool = _midgame_env(torch, device, n)
    stream = torch.cuda.current_stream(device)
    table = (ctypes.c_void_p * K)(*[pool[t & 7].data_ptr() for t in range(K)])
    out = {'workload': workload_text(n) + f'; one env2048_step_sequence call of K = {K} steps', 'K': K}
    direct_here = False if stream_only else lib.direct_stats(device_index)['available']  # (the query would start the route: not in a stream-only pass)
    # (the stream route's legs come from stream_route_legs, run before the library's queues exist -- see there)
    legs = [('default', 0)] if direct_here and not stream_only else []
    if stream_only or not direct_here:
        legs += [('two_chains_default', capi.FLAG_NO_DIRECT), ('single_chain', capi.FLAG_NO_DIRECT | capi.FLAG_SINGLE_CHAIN)]
    for k, (name, extra) in enumerate(legs):
        flags = env.flags | extra
        calls_before = 0 if stream_only else lib.direct_stats(device_index)['calls']
        wall1, ev1 = _timed_sequence(torch, lib, stream, device_index, h, env, table, K, flags, 1000 + 100 * K * k)
        wall, ev = _timed_sequence(torch, lib, stream, device_index, h, env, table, K, flags, 50_000 + 100 * K * k, calls=5)
        info = {'calls': 0, 'last_chains': 0, 'last_gpu_ns': 0} if stream_only else lib.direct_stats(device_index)
        took_direct = info['calls'] > calls_before
        gbs = ALGO_BYTES_PER_STEP * n / (ev / K) / 1e9
        out[name] = {'route': 'direct (the library\'s own queues, boards L2-resident between the first and the last launch of a call)'
                              if took_direct else 'stream (HIP launches)',
                     'launch_plan': dict(lib.step_sequence_plan(n, K, flags), **({'chains': info['last_chains'], 'write_through': False} if took_direct else {})),
                     'device_us_per_step_last_call': info['last_gpu_ns'] / K * 1e-3 if took_direct and info['last_gpu_ns'] else None,
                     'event_us_per_step': ev / K * 1e6, 'wall_us_per_step': wall / K * 1e6,
                     'value': n * K / wall, 'unit': 'board-steps/s', 'timed_region': f'5 consecutive calls of K = {K} steps between two synchronisations',
                     'one_call_between_synchronisations': {'event_us_per_step': ev1 / K * 1e6, 'wall_us_per_step': wall1 / K * 1e6,
                                                           'note': 'a single 1.5 ms call from an idle GPU: fork / join of the second chain '
                                                                   'and the clock ramp are not amortised'},
                     'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                  'regime': 'cache-resident equivalent (algorithmic bytes / event time per step; see roofline.regime)'}}
    if stream_side and 'configs[2]_long_sequence' in stream_side:  # the stream route's legs, measured by the child process
        for key in ('two_chains_default', 'single_chain', 'two_chains_speedup', 'helper_thread'):
            out[key] = stream_side['configs[2]_long_sequence'].get(key)
        out['stream_route_measured_in'] = stream_side.get('where')
    if out.get('single_chain') and out.get('two_chains_default'):
        out['two_chains_speedup'] = out['single_chain']['event_us_per_step'] / out['two_chains_default']['event_us_per_step']
    if out.get('default') and out.get('two_chains_default'):
        out['direct_speedup_over_two_stream_chains'] = out['two_chains_default']['wall_us_per_step'] / out['default']['wall_us_per_step']
    if 'two_chains_default' not in [name for name, _ in legs]:
        return out
    # the helper thread that enqueues the second chain: CPU time it consumed over a K = 2000 call (its poll window included)
    K2 = 2000
    table2 = (ctypes.c_void_p * K2)(*[pool[t & 7].data_ptr() for t in range(K2)])
    before = lib.helper_stats(device_index)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.step_sequence(device_index, h, n, env.env_base, env.boards.data_ptr(), table2, env.reward.data_ptr(), env.done.data_ptr(), None, None, 0,
                      SEED, 50_000, K2, env.flags | capi.FLAG_NO_DIRECT)
    enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall2 = time.perf_counter() - t0
    time.sleep(0.002)  # let the helper's poll window (default 300 us) run out so that its cost is in the sample
    after = lib.helper_stats(device_index)
    out['helper_thread'] = {
        'K': K2, 'wall_ms': wall2 * 1e3, 'caller_enqueue_ms': enqueued * 1e3, 'us_per_step_wall': wall2 / K2 * 1e6,
        'helper_cpu_ms': (after['cpu_ns'] - before['cpu_ns']) * 1e-6, 'helper_jobs': after['jobs'] - before['jobs'], 'spin_us': after['spin_us'],
        'note': 'CPU time of the library\'s helper thread (CLOCK_THREAD_CPUTIME_ID, env2048_helper_stats) over one two-chain call of K steps: it '
                'enqueues the second chain (K launches) and then polls for ENV2048_HELPER_SPIN_US before sleeping; one hardware thread per '
                'device while a stepping loop is running, nothing while idle (include/env2048.h, "The helper thread")'}
    return out


def legal_obs_leg(torch, lib, device, device_index, h, n: int = 1 << 20, K: int = 200, stream_side=None, stream_only: bool = False) -> dict:
    """What the reference's caller needs EVERY move (reinforce/training/self_play.py:316-317,354): transition + legal mask +
    observation of the new boards, from the EXTRAS variant of the step kernel in the same launch -- with float32 and with
    bfloat16 observations (the same values: e/16 is exact in both)."""
    out = {'workload': workload_text(n) + f'; legal mask + observation written by the same launch; one env2048_step_sequence call of K = {K} steps',
           'K': K, 'replaces': 'batched_step + batched_get_observation + batched_get_legal_actions, reinforce/game/env.py:193-245'}
    stream = torch.cuda.current_stream(device)
    for name, dtype, obs_bytes in (('obs_f32', torch.float32, 4), ('obs_bf16', torch.bfloat16, 2)):
        env, pool = _midgame_env(torch, device, n, with_legal=True, with_obs=True, obs_dtype=dtype)
        table = (ctypes.c_void_p * K)(*[pool[t & 7].data_ptr() for t in range(K)])
        from simulate_2048_amd import capi

        flags = env.flags | (capi.FLAG_NO_DIRECT if stream_only else 0)
        calls_before = 0 if stream_only else lib.direct_stats(device_index)['calls']
        wall, ev = _timed_sequence(torch, lib, stream, device_index, h, env, table, K, flags, 1000, legal=env.legal.data_ptr(),
                                   obs=env.obs.data_ptr())
        took_direct = (not stream_only) and lib.direct_stats(device_index)['calls'] > calls_before
        per_board = ALGO_BYTES_PER_STEP + 1 + 16 * obs_bytes
        gbs = per_board * n / (ev / K) / 1e9
        plan = dict(lib.step_sequence_plan(n, K, flags), **({'route': 'direct', 'chains': 2, 'write_through': False} if took_direct else {'route': 'stream'}))
        out[name] = {'launch_plan': plan, 'event_us_per_step': ev / K * 1e6, 'wall_us_per_step': wall / K * 1e6, 'value': n * K / wall,
                     'unit': 'board-steps/s',
                     'on_the_stream_route': None if not stream_side else (stream_side.get('configs[2]_with_legal_obs') or {}).get(name),
                     'roofline': {'bound': 'hbm', 'kernel': (f'env2048_step_resident_large_extras' if took_direct else
                                                             f'step_kernel<true, false, {plan["boards_per_lane"]}>'),
                                  'algorithmic_bytes_per_board_step': per_board, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': gbs / HBM_PEAK_GBS,
                                  'regime': ('L2-resident boards, observation rows written through: an equivalent (algorithmic bytes / event time per '
                                             'step) -- what crosses the fabric per board-step is the action, the reward / done / legal bytes and the '
                                             f'{16 * obs_bytes} observation bytes' if took_direct else
                                             f'cache-resident equivalent: {n * (16 + 8 + 4 + 1 + 1 + 16 * obs_bytes) / 2**20:.0f} MiB working set < 256 MiB '
                                             f'Infinity Cache (algorithmic bytes / event time per step)')}}
        del env, pool
    return out


def stream_route_legs(torch, lib, device, device_index, h) -> dict:
    """The legs of other_configs on the STREAM route (ENV2048_FLAG_NO_DIRECT: HIP launches) -- configs[1] as 2000 launches,
    configs[4] as 2 x 512 launches, configs[2] as one K = 200 sequence (two chains and one) with the helper thread's cost.  Run by
    run_rank BEFORE anything else of the run: the two-chain stream sequence needs its two HIP streams on two hardware pipes, and
    a process that has created more than four HIP streams (the later legs of this file do) puts two of them on one (11 us per
    step instead of 6.6-7, with or without the library's own queues: tools/diag_pipe_sharing.py, profiles/r05_w_*)."""
    from simulate_2048_amd import capi
    from simulate_2048_amd.batch import BatchedEnv2048

    stream = torch.cuda.current_stream(device)
    no_direct = capi.FLAG_NO_DIRECT
    out = {'where': 'this process, before the direct route was started (ENV2048_FLAG_NO_DIRECT on every call; the library had created no queue yet)'}
    # these legs are the first GPU work of the process: bring the clocks up first (0.1 s of 2^20-board steps), as the headline's
    # own blocks do for the legs that follow it -- the launch-bound 65 536-board leg measured 4.5 instead of 3.2 us per step cold
    warm, _ = _midgame_env(torch, device, 1 << 20)
    lib.step_sequence(device_index, h, warm.num_envs, 0, warm.boards.data_ptr(), None, warm.reward.data_ptr(), warm.done.data_ptr(), None, None, 0, SEED,
                      1 << 40, 12000, warm.flags | capi.FLAG_RANDOM_ACTIONS | no_direct | capi.FLAG_SINGLE_CHAIN)
    torch.cuda.synchronize()
    del warm

    def timed(call, reps=4):
        times = []
        for rep in range(reps):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record(stream)
            call(rep)
            ev1.record(stream)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
        return sorted(times)[len(times) // 2]

    n1, count = 1 << 16, 2000
    env, pool = _midgame_env(torch, device, n1)
    table = (ctypes.c_void_p * count)(*[pool[t & 7].data_ptr() for t in range(count)])
    wall, ev = timed(lambda rep: lib.step_sequence(device_index, h, n1, 0, env.boards.data_ptr(), table, env.reward.data_ptr(), env.done.data_ptr(),
                                                   None, None, 0, SEED, 64 + rep * count, count, env.flags | no_direct))
    out['configs[1]'] = {'value': n1 * count / wall, 'unit': 'board-steps/s', 'us_per_step_wall': wall / count * 1e6, 'us_per_step_event': ev / count * 1e6,
                         'note': 'the same call on the stream route: 2000 HIP launches'}
    del env, pool
    n4, T = 1 << 20, 512
    a = BatchedEnv2048(n4, SEED, device=device, auto_reset=True)
    a.reset()
    start = a.boards.clone()
    reward_rows = torch.zeros((T, n4), dtype=torch.int32, device=device)
    done_rows = torch.zeros((T, n4), dtype=torch.uint8, device=device)

    def rows_call(rep):
        lib.step_sequence(device_index, h, n4, 0, a.boards.data_ptr(), None, reward_rows.data_ptr(), done_rows.data_ptr(), None, None, n4, SEED, 0, T,
                          capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS | no_direct)

    rows_call(0)  # (pays for the second chain's stream / helper thread)
    times = []
    for rep in range(3):
        a.boards.copy_(start)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        rows_call(rep)
        ev1.record(stream)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3))
    wall, ev = sorted(times)[1]
    out['configs[4]'] = {'value': n4 * T / wall, 'unit': 'board-steps/s', 'ms_total_wall': wall * 1e3, 'us_per_step_event': ev / T * 1e6,
                         'note': 'the same call on the stream route: 2 x 512 HIP launches (two streams, helper thread)'}
    del a, start, reward_rows, done_rows
    out['configs[2]_long_sequence'] = long_sequence_leg(torch, lib, device, device_index, h, stream_only=True)
    out['configs[2]_with_legal_obs'] = legal_obs_leg(torch, lib, device, device_index, h, stream_only=True)
    return out


def assemble_single_process_result(args, walls, per_device_ms, roofline: dict) -> dict:
    """The JSON object of a --single-process run: same contract, value = boards x K x devices / the median block's wall time."""
    G, n, K, W = args.gpus, args.boards, args.steps, args.warmup
    wall = statistics.median(walls)
    out = {
        'metric': 'env_board_steps_per_sec', 'value': float(n) * G * K / wall, 'unit': 'board-steps/s', 'n_gpus': G, 'steps': K, 'warmup': W,
        'ms_per_step': wall * 1e3 / K, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': workload_text(n), 'boards_per_gpu': n, 'global_boards': n * G,
                   'launch_mode': 'single process: ShardedEnv2048, one worker thread + one stream per device, each enqueuing its K launches '
                                  'through one env2048_step_sequence call',
                   'sharding': 'contiguous env-index slices, no collective, no torch.distributed', 'outputs': 'board,reward,done'},
        'timing': {'blocks': len(walls), 'block_steps': K,
                   'wall_us_per_step': {'median': wall / K * 1e6, 'min': min(walls) / K * 1e6, 'max': max(walls) / K * 1e6},
                   'per_device_event_us_per_step': [m * 1e3 / K for m in per_device_ms]},
        'roofline': roofline,
        'cpu_baseline': None, 'cpu_baseline_note': 'timed by the default (rank-per-GPU, N = 1) run only',
    }
    if args.shards_on_device_0:
        out['config']['rehearsal'] = f'{G} shards SHARING device 0: exercises the single-process fan-out only; not a scaling measurement'
    return out


def run_single_process(args) -> None:
    """--single-process: the whole node from ONE process (ShardedEnv2048: contiguous slices, one worker thread and one stream
    per device, no collective, no torch.distributed).  A timed block = synchronise all devices | every shard's K launches
    enqueued by its own thread through one env2048_step_sequence call | synchronise all devices."""
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch

    from simulate_2048_amd import capi
    from simulate_2048_amd.sharded import ShardedEnv2048

    if not torch.cuda.is_available():
        sys.exit('bench.py needs a HIP device: the product path has no CPU fallback')
    G, n, K, W = args.gpus, args.boards, args.steps, args.warmup
    rehearsal = args.shards_on_device_0
    if not rehearsal and G > torch.cuda.device_count():
        sys.exit(f'--gpus {G} but only {torch.cuda.device_count()} visible device(s); --shards-on-device-0 rehearses the fan-out on one')
    devices = [torch.device('cuda', 0 if rehearsal else g) for g in range(G)]
    env = ShardedEnv2048(n * G, SEED, devices=devices, auto_reset=True)
    lib = capi.library()
    env.reset()
    env.step_sequence(64)  # mid-game boards (on-device random actions)
    gen = torch.Generator(device='cpu').manual_seed(1)
    pools = [[torch.randint(0, 4, (c,), generator=gen, dtype=torch.uint8).to(d) for _ in range(8)] for (_, c), d in zip(env.ranges, devices)]
    tables = {}

    def tables_for(count: int, first: int):
        key = (count, first & 7)
        if key not in tables:
            tables[key] = [(ctypes.c_void_p * count)(*[pool[(first + t) & 7].data_ptr() for t in range(count)]) for pool in pools]
        return tables[key]

    for phase in range(8):
        tables_for(K, phase)
    env.step_sequence(W, tables_for(W, 0)) if W else None
    env.synchronize()

    def timed_block():
        env.synchronize()
        t0 = time.perf_counter()
        env.step_sequence(K, tables_for(K, env.step_index))
        env.synchronize()
        return time.perf_counter() - t0

    calib = timed_block()
    blocks = int(min(args.max_blocks, max(3, math.ceil(args.min_ms * 1e-3 / max(calib, 1e-7)))))
    walls = [timed_block() for _ in range(blocks)]
    per_device_ms = env.timed(lambda g: env.shards[g].lib.step_sequence(
        g if not rehearsal else 0, torch.cuda.current_stream(devices[g]).cuda_stream, env.shards[g].num_envs, env.shards[g].env_base,
        env.shards[g].boards.data_ptr(), tables_for(K, 0)[g], env.shards[g].reward.data_ptr(), env.shards[g].done.data_ptr(), None, None, 0,
        SEED, 10_000_000, K, env.shards[g].flags))
    launch_s = statistics.median(per_device_ms) * 1e-3 / K
    info = lib.direct_stats(0)
    route = {'direct': info['calls'] > 0, 'chains': info['last_chains'], 'reruns': info['reruns'], 'state': info['reason'], 'device_s_per_step': None}
    out = assemble_single_process_result(args, walls, per_device_ms, build_roofline(args, lib, n, env.shards[0].flags, launch_s, 4, route=route))
    env.close()
    os.write(result_fd, (json.dumps(out) + '\n').encode())


def main():
    args = parse_args()
    if args.single_process:
        run_single_process(args)
        return
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no torchrun environment: start one worker per GPU ourselves (children; this process never touches a GPU)
        sys.exit(spawn_ranks(args.gpus, [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:]))
    run_rank(args)


if __name__ == '__main__':
    main()
