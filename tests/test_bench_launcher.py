"""
CPU tests of bench.py's own process launcher (`python bench.py --gpus N` without a torchrun environment) with stub
workers, of its profile/roofline helpers, and of the multi-launch entry points through the oracle-backed stand-in.
"""

import ctypes
import json
import subprocess
import sys
import textwrap

import numpy as np
import torch

import oracle
from conftest import REPO
from oracle_backend import OracleLibrary
from simulate_2048_amd import capi

sys.path.insert(0, str(REPO))
import bench  # noqa: E402


def _worker(tmp_path, body: str):
    path = tmp_path / 'stub_worker.py'
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def test_launcher_spawns_one_rank_per_gpu_and_relays_rank0(tmp_path, capfd):
    """Two gloo ranks rendezvous through the environment the launcher set up, rank 0's JSON line is relayed."""
    cmd = _worker(tmp_path, '''
        import json, os, torch, torch.distributed as dist
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        assert rank == int(os.environ["RANK"]) == int(os.environ["LOCAL_RANK"]) and world == int(os.environ["WORLD_SIZE"]) == 2
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        print("noise from rank", rank)          # only rank 0's stdout is relayed
        if rank == 0:
            print(json.dumps({"n_gpus": world, "sum": float(t)}))
        dist.destroy_process_group()
    ''')
    rc = bench.spawn_ranks(2, cmd, timeout=240)
    out = capfd.readouterr().out
    assert rc == 0, out
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == {'n_gpus': 2, 'sum': 3.0}
    assert 'noise from rank 1' not in out


def test_launcher_fails_when_a_rank_fails_and_stops_the_others(tmp_path, capfd):
    cmd = _worker(tmp_path, '''
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)       # rank 0 would hang forever waiting for its peer
    ''')
    rc = bench.spawn_ranks(2, cmd, timeout=120)
    assert rc == 3
    assert 'rank 1 exited with code 3' in capfd.readouterr().err


def test_launcher_requires_a_json_line(tmp_path, capfd):
    rc = bench.spawn_ranks(2, _worker(tmp_path, 'print("no json here")'), timeout=60)
    assert rc == 1
    assert 'no JSON line' in capfd.readouterr().err


def test_bench_without_gpu_exits_nonzero_in_every_mode():
    """No GPU in this container: the single-rank run and the self-spawned 2-rank run both fail loudly (no CPU fallback)."""
    if torch.cuda.is_available():
        return
    for extra in ([], ['--gpus', '2', '--dist-backend', 'gloo']):
        proc = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--steps', '2', '--warmup', '1'] + extra, capture_output=True,
                              text=True, timeout=300)
        assert proc.returncode != 0
        assert 'no CPU fallback' in proc.stderr
        assert not any(l.startswith('{') for l in proc.stdout.splitlines())


def test_profile_table_is_keyed_by_kernel_source():
    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and sha == bench.kernel_source_sha()
    # the hash covers the step kernel's own source regions (device code + host launch policy) and the headers they use
    import re

    text = (REPO / 'simulate_2048_amd' / 'csrc' / 'step_kernels.hip').read_text()
    regions = re.findall(r'\[step-kernel-source-begin\](.*?)\[step-kernel-source-end\]', text, flags=re.S)
    assert len(regions) == 3 and 'void step_kernel(' in regions[0] and 'void launch_step(' in regions[1] and 'env2048_step_resident_large' in regions[2]
    assert 'rollout_kernel' not in ''.join(regions) and 'expectimax_ply_kernel' not in ''.join(regions)
    path = REPO / 'profiles' / 'step_pmc.json'
    if path.exists():
        table = json.loads(path.read_text())
        assert set(table) >= {'kernel_source_sha', 'entries'}
        for key, entry in table['entries'].items():
            leg, boards = key.split(':')
            assert leg in ('direct', 'resident', 'resident_one_chain', 'cold', 'cold_separate') and int(boards) > 0
            assert entry['fabric_bytes_per_launch'] > 0 and (REPO / entry['source']).exists()
        prof = bench.load_step_profile(1 << 20, 'resident')
        assert prof is None or prof['stale'] == (table['kernel_source_sha'] != sha)
    assert bench.load_step_profile(12345, 'resident') is None
    text = bench.workload_text(1 << 20)
    assert text.startswith('BASELINE configs[2]: 1 048 576 parallel') and bench.workload_text(1 << 16).startswith('BASELINE configs[1]: 65 536')
    assert 'non-default size: 4 194 304' in bench.workload_text(1 << 22)


def test_step_sequence_and_step_many_host_logic():
    """The record / row arithmetic of the multi-launch entry points, through the oracle-backed stand-in."""
    lib = OracleLibrary()
    n, T, stride, seed = 300, 6, 320, 11
    start, _ = oracle.reset_batch(n, seed, 0, env_base=8)
    rng = np.random.default_rng(2)
    acts = [np.ascontiguousarray(rng.integers(0, 4, size=n).astype(np.uint8)) for _ in range(T)]
    table = (ctypes.c_void_p * T)(*[a.ctypes.data for a in acts])
    boards = start.copy()
    reward = np.full((T, stride), -3, dtype=np.int32)
    done = np.full((T, stride), 9, dtype=np.uint8)
    obs = np.full((T, stride, 16), -1, dtype=np.float32)
    lib.step_sequence(0, None, n, 8, boards.ctypes.data, table, reward.ctypes.data, done.ctypes.data, None, obs.ctypes.data, stride,
                      seed, 100, T, capi.FLAG_AUTO_RESET)
    ref = start
    for t in range(T):
        r = oracle.step_batch(ref, acts[t], seed, 100 + t, env_base=8, flags=1, want_obs=True)
        ref = r['boards']
        np.testing.assert_array_equal(reward[t, :n], r['reward'])
        np.testing.assert_array_equal(done[t, :n], r['done'])
        np.testing.assert_array_equal(obs[t, :n], r['obs'])
    np.testing.assert_array_equal(boards, ref)
    assert (reward[:, n:] == -3).all() and (done[:, n:] == 9).all()
    # step_many: two independent batches in one call
    b1, b2 = start.copy(), start[:100].copy()
    d1, d2 = np.zeros(n, np.uint8), np.zeros(100, np.uint8)
    records = (capi.StepArgs * 2)(
        capi.StepArgs(n, 8, b1.ctypes.data, acts[0].ctypes.data, None, d1.ctypes.data, None, None, seed, 5, 1, 0),
        capi.StepArgs(100, 8, b2.ctypes.data, None, None, d2.ctypes.data, None, None, seed, 5, 3, 0))
    lib.step_many(0, None, records)
    np.testing.assert_array_equal(b1, oracle.step_batch(start, acts[0], seed, 5, env_base=8, flags=1)['boards'])
    np.testing.assert_array_equal(b2, oracle.step_batch(start[:100], None, seed, 5, env_base=8, flags=3)['boards'])
    assert ctypes.sizeof(capi.StepArgs) == 104  # the C struct of include/env2048.h: 2 u64 + 6 pointers + 2 u64 + 2 u32 + 2 pointers
    assert ctypes.sizeof(capi.Episode) == 24


# ---- multi-GPU readiness without the hardware ------------------------------------------------------------
def _validate_contract_line(out: dict, n_gpus: int) -> None:
    """The driver's contract for one bench line (what SCALE_rNN.json is computed from) at N = n_gpus."""
    assert out['metric'] == 'env_board_steps_per_sec' and out['unit'] == 'board-steps/s' and out['higher_is_better'] is True
    assert out['n_gpus'] == n_gpus and out['scaling'] == 'weak' and out['vs_baseline'] is None and out['dtype'] == 'u8' and out['data'] == 'synthetic'
    assert out['config']['boards_per_gpu'] == 1 << 20 and out['config']['global_boards'] == n_gpus << 20
    assert out['config']['workload'].startswith('BASELINE configs[2]') and 'model' not in out['config']
    assert out['cpu_baseline'] is None and 'N = 1' in out['cpu_baseline_note']
    assert out['roofline']['bound'] == 'hbm' and out['roofline']['peak'] == 8000.0 and out['roofline']['unit'] == 'GB/s'
    assert abs(out['roofline']['frac'] - out['roofline']['achieved'] / 8000.0) < 1e-12


def test_eight_ranks_reduce_their_timings_and_emit_the_contract_line(tmp_path, capfd):
    """`python bench.py --gpus 8` without the GPUs: eight gloo ranks started by bench.py's own launcher run the benchmark's
    own reduction code (RankComm: barrier, max over ranks per block, gather of the per-rank medians) on synthetic block times
    -- rank r is (1 + r/100) times slower than rank 0 -- and rank 0 assembles the line with bench.assemble_result.
    value must be the WHOLE-JOB rate at the pace of the slowest rank; global_boards = 8 x 2^20 = BASELINE configs[3]."""
    cmd = _worker(tmp_path, f'''
        import json, os, sys
        sys.path.insert(0, {str(REPO)!r})
        import bench
        args = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo"])
        rank = int(os.environ["RANK"])
        comm = bench.RankComm("gloo")
        assert comm.world == 8
        comm.barrier()
        slow = 1.0 + rank / 100.0
        walls = [20 * 8.8e-6 * slow * (1.0 + 0.01 * b) for b in range(7)]      # seconds per 20-step block
        events = [20 * 7.9e-6 * slow * (1.0 + 0.01 * b) for b in range(7)]
        timings = bench.reduce_timings(comm, walls, events, args.steps)
        if rank == 0:
            roofline = dict(bound="hbm", achieved=38 * 2**20 / timings["launch_s"] / 1e9, peak=8000.0, unit="GB/s")
            roofline["frac"] = roofline["achieved"] / roofline["peak"]
            print(json.dumps(bench.assemble_result(args, comm.world, args.boards, timings, 7, 3.5, roofline)))
        comm.teardown()
    ''')
    rc = bench.spawn_ranks(8, cmd, timeout=600)
    captured = capfd.readouterr()
    assert rc == 0, captured.out + captured.err
    lines = [l for l in captured.out.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    _validate_contract_line(out, 8)
    assert out['steps'] == 20 and out['warmup'] == 5
    per_rank = out['timing']['per_rank_us_per_step']
    assert [e['rank'] for e in per_rank] == list(range(8))
    assert all(abs(e['event'] - 7.9 * (1 + r / 100) * 1.03) < 1e-6 for r, e in enumerate(per_rank))  # each rank's own median block
    slowest_wall = 20 * 8.8e-6 * 1.07 * 1.03  # median block of the slowest rank
    assert abs(out['ms_per_step'] - slowest_wall * 1e3 / 20) < 1e-9
    assert abs(out['value'] - 8 * 2**20 * 20 / slowest_wall) / out['value'] < 1e-9
    assert abs(out['timing']['event_us_per_step']['median'] - 7.9 * 1.07 * 1.03) < 1e-6


def test_single_process_line_for_eight_devices():
    """`bench.py --single-process --gpus 8`: the same contract from ONE process (ShardedEnv2048; its fan-out over 8 devices is
    covered by tests/test_sharded.py) -- the line assembled from synthetic per-device times."""
    args = bench.parse_args(['--single-process', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    walls = [20 * 9.0e-6 * (1 + 0.02 * b) for b in range(9)]
    roofline = dict(bound='hbm', achieved=5000.0, peak=8000.0, unit='GB/s', frac=5000.0 / 8000.0)
    out = bench.assemble_single_process_result(args, walls, [0.16 + 0.001 * g for g in range(8)], roofline)
    _validate_contract_line(out, 8)
    assert len(out['timing']['per_device_event_us_per_step']) == 8 and out['timing']['blocks'] == 9
    median_wall = sorted(walls)[4]
    assert abs(out['value'] - 8 * 2**20 * 20 / median_wall) / out['value'] < 1e-12
    assert 'rehearsal' not in out['config']
    args.shards_on_device_0 = True
    assert 'not a scaling measurement' in bench.assemble_single_process_result(args, walls, [0.16] * 8, roofline)['config']['rehearsal']


def test_roofline_object_is_the_hbm_fraction_with_the_valu_analysis_beside_it():
    """top-level roofline = algorithmic bytes / launch time against 8 TB/s (labelled cache-resident at 2^20
    boards); the instruction-issue analysis sits under roofline.valu; counters come from the profile of the TIMED shape
    (one chain for the driver's --steps 20, two chains from K = 32)."""

    class Lib:
        def step_sequence_plan(self, n, count, flags):
            return capi.Library().step_sequence_plan(n, count, flags)

    for steps, leg, chains in ((20, 'resident_one_chain', 1), (200, 'resident', 2)):
        args = bench.parse_args(['--steps', str(steps)])
        r = bench.build_roofline(args, Lib(), 1 << 20, capi.FLAG_AUTO_RESET, 8.0e-6, 4)
        assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
        assert abs(r['achieved'] - 38 * 2**20 / 8.0e-6 / 1e9) < 1e-6 and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-12
        assert 'cache-resident' in r['regime'] and r['launch_plan']['chains'] == chains and r['launches_per_step'] == chains
        assert r['algorithmic_bytes_per_launch'] == 38 << 20
        if r['traffic_source'] is not None:  # a committed counter table: it must be the table of the timed shape
            assert r['traffic_source']['leg'] == leg and 0.9 < r['traffic'] / r['algorithmic_bytes_per_launch'] < 1.3
        if 'valu' in r:
            assert r['valu']['unit'] == 'G wave-instr/s' and 0 < r['valu']['frac'] < 1
    big = bench.build_roofline(bench.parse_args(['--boards', str(1 << 24)]), Lib(), 1 << 24, capi.FLAG_AUTO_RESET, 1.0e-4, 4)
    assert big['regime'].startswith('HBM') and 'valu' not in big
