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
    assert len(regions) == 2 and 'void step_kernel(' in regions[0] and 'void launch_step(' in regions[1]
    assert 'rollout_kernel' not in ''.join(regions) and 'expectimax_ply_kernel' not in ''.join(regions)
    path = REPO / 'profiles' / 'step_pmc.json'
    if path.exists():
        table = json.loads(path.read_text())
        assert set(table) >= {'kernel_source_sha', 'entries'}
        for key, entry in table['entries'].items():
            leg, boards = key.split(':')
            assert leg in ('resident', 'cold', 'cold_separate') and int(boards) > 0
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
