"""
Host-side logic that needs no GPU: the C-ABI surface of the built library, loud failure without a
device, BatchedEnv2048 buffer/step bookkeeping (through the oracle-backed stand-in), sharding
arithmetic and a 2-process gloo run of the sharded path.
"""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import REPO
from oracle_backend import OracleLibrary
from simulate_2048_amd import BatchedEnv2048, capi
from simulate_2048_amd.sharding import shard_range


def test_library_exports_every_declared_symbol():
    header = (REPO / 'include' / 'env2048.h').read_text()
    declared = set(re.findall(r'\b(env2048_[a-z0-9_]+)\s*\(', header))
    assert declared == set(capi.PROTOTYPES), declared ^ set(capi.PROTOTYPES)
    dll = ctypes.CDLL(str(capi.LIB_PATH))
    for name in declared:
        assert hasattr(dll, name), name
    assert capi.Library().abi_version() == capi.ABI_VERSION
    assert f'#define ENV2048_ABI_VERSION {capi.ABI_VERSION}' in header


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C-ABI: include/env2048.h compiles as strict C99 and as C++, and a C program linked
    against libenv2048.so reaches the non-compute entry points (argument validation needs no GPU)."""
    src = tmp_path / 'abi.c'
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "env2048.h"\n'
        'int main(void) {\n'
        '  if (env2048_abi_version() != ENV2048_ABI_VERSION) return 1;\n'
        '  /* a NULL board pointer is refused before any device is touched */\n'
        '  int rc = env2048_step(0, NULL, 4, 0, NULL, NULL, NULL, NULL, NULL, NULL, 1, 0, ENV2048_FLAG_AUTO_RESET);\n'
        '  if (rc != ENV2048_ERR_INVALID_ARG) return 2;\n'
        '  if (strlen(env2048_last_error()) == 0) return 3;\n'
        '  if (env2048_step(0, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, 1, 0, 0) != ENV2048_OK) return 4; /* n = 0: no-op */\n'
        '  puts("ok");\n  return 0;\n}\n')
    inc = str(REPO / 'include')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', f'-I{inc}', '-fsyntax-only', str(src)], check=True)
    subprocess.run(['g++', '-std=c++11', '-Wall', '-Wextra', '-pedantic', '-Werror', f'-I{inc}', '-x', 'c++', '-fsyntax-only', str(src)],
                   check=True)
    exe = tmp_path / 'abi'
    lib_dir = str(capi.LIB_PATH.parent)
    subprocess.run(['gcc', '-std=c99', f'-I{inc}', str(src), f'-L{lib_dir}', '-lenv2048', f'-Wl,-rpath,{lib_dir}', '-o', str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', (out.returncode, out.stdout, out.stderr)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the behaviour on a machine WITHOUT a GPU')
def test_product_fails_loudly_without_gpu():
    lib = capi.Library()
    with pytest.raises(capi.Env2048Error) as e:
        lib.device_count()
    assert e.value.code == capi.ERR_DEVICE
    buf = np.zeros(64, dtype=np.uint8)
    ptr = (buf.ctypes.data + 15) // 16 * 16
    with pytest.raises(capi.Env2048Error) as e:  # valid arguments, no device -> ERR_DEVICE, never a CPU result
        lib.step(0, None, 1, 0, ptr, ptr + 16, None, ptr + 32, None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_DEVICE and 'no CPU fallback' in str(e.value)
    with pytest.raises(capi.Env2048Error) as e:  # argument errors are reported before the device is touched
        lib.step(0, None, 1, 0, None, None, None, None, None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.step(0, None, 1, 0, ptr + 1, ptr + 16, None, ptr + 32, None, None, 0, 0, 0)
    assert e.value.code == capi.ERR_ALIGNMENT
    with pytest.raises(RuntimeError):
        BatchedEnv2048(8, device='cuda')  # no HIP device
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        BatchedEnv2048(8, device='cpu')  # CPU devices are refused outright
    from simulate_2048_amd import runtime

    with pytest.raises((RuntimeError, capi.Env2048Error)):
        runtime.current()
    lib.step(0, None, 0, 0, None, None, None, None, None, None, 0, 0, 0)  # n = 0 stays a no-op


def test_missing_library_raises(tmp_path):
    with pytest.raises(FileNotFoundError, match='no CPU fallback'):
        capi.Library(tmp_path / 'libenv2048.so')


def test_product_package_never_imports_the_oracle():
    code = 'import sys, simulate_2048_amd, simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight, ' \
           'simulate_2048_amd.reinforce.game.env, simulate_2048_amd.sharding; ' \
           'assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), "oracle imported"'
    subprocess.run([sys.executable, '-c', code], check=True, cwd=REPO)
    for path in (REPO / 'simulate_2048_amd').rglob('*.py'):
        text = path.read_text()
        assert not re.search(r'^\s*(import oracle|from oracle)', text, flags=re.M), path


def test_batched_env_bookkeeping():
    lib = OracleLibrary()
    env = BatchedEnv2048(300, seed=9, device='cpu', env_base=40, auto_reset=True, with_legal=True, with_obs=True, lib=lib)
    boards = env.reset()
    ref, _ = oracle.reset_batch(300, 9, 0, env_base=40)
    np.testing.assert_array_equal(boards.numpy(), ref)
    np.testing.assert_array_equal(env.legal.numpy(), oracle.status_batch(ref)['legal'])
    rng = np.random.default_rng(0)
    for t in range(30):
        actions = torch.from_numpy(rng.integers(0, 4, size=300).astype(np.int64))  # wrong dtype on purpose: converted
        b, r, d = env.step(actions)
        res = oracle.step_batch(ref, actions.numpy().astype(np.uint8), 9, t, env_base=40, flags=1, want_legal=True, want_obs=True)
        ref = res['boards']
        np.testing.assert_array_equal(b.numpy(), ref)
        np.testing.assert_array_equal(r.numpy(), res['reward'])
        np.testing.assert_array_equal(env.obs.numpy(), res['obs'])
    assert env.step_index == 30 and b.data_ptr() == env.boards.data_ptr()  # persistent buffers, in place
    env.step(None)  # on-device random actions
    ref = oracle.step_batch(ref, None, 9, 30, env_base=40, flags=3)['boards']
    np.testing.assert_array_equal(env.boards.numpy(), ref)
    out = env.rollout(50)
    ro = oracle.rollout_batch(ref, 9, 31, 50, env_base=40, flags=3)
    np.testing.assert_array_equal(env.boards.numpy(), ro['boards'])
    np.testing.assert_array_equal(out['return_sum'].numpy(), ro['return_sum'])
    assert env.step_index == 81
    with pytest.raises(ValueError):
        env.step(torch.zeros(5, dtype=torch.uint8))
    st = env.status()
    np.testing.assert_array_equal(st['n_empty'].numpy(), (ro['boards'] == 0).sum(axis=1))
    assert [c[0] for c in lib.calls[:3]] == ['reset', 'status', 'step']


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_batched_env_sixteen_bit_observations(dtype):
    """obs_dtype: same values (exact), initial observation after reset included, rollout unaffected."""
    lib = OracleLibrary()
    env = BatchedEnv2048(130, seed=3, device='cpu', auto_reset=True, with_legal=True, with_obs=True, obs_dtype=dtype, lib=lib)
    ref = env.reset().numpy().copy()
    assert env.obs.dtype == dtype
    np.testing.assert_array_equal(env.obs.to(torch.float32).numpy(), oracle.status_batch(ref)['obs'])
    for t in range(20):
        env.step(None)
        res = oracle.step_batch(ref, None, 3, t, flags=3, want_obs=True)
        ref = res['boards']
        np.testing.assert_array_equal(env.obs.to(torch.float32).numpy(), res['obs'])
    env.rollout(5)
    with pytest.raises(ValueError):
        BatchedEnv2048(4, device='cpu', with_obs=True, obs_dtype=torch.float64, lib=lib)
    with pytest.raises(ValueError):
        env.step(None, out_obs=torch.zeros((130, 16), dtype=torch.float32))  # wrong dtype for this env


def test_shard_range_covers_everything_once():
    for total in (0, 1, 5, 1024, (1 << 20) + 3, 8 << 20):
        for world in (1, 2, 3, 4, 8):
            slices = [shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in slices) == total
            pos = 0
            for base, count in slices:
                assert base == pos or count == 0
                assert base % 4 == 0 or count == 0
                pos += count
    assert shard_range(8 << 20, 3, 8) == (3 << 20, 1 << 20)  # BASELINE config 4: 8 x 2^20
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_sharded_run_equals_single_run():
    """Slices with env_base = slice start reproduce the unsliced batch exactly (host logic, 3 'ranks')."""
    total, seed, T = 1000, 77, 25
    whole = BatchedEnv2048(total, seed, device='cpu', auto_reset=True, lib=OracleLibrary())
    whole.reset()
    parts = []
    for r in range(3):
        base, count = shard_range(total, r, 3)
        e = BatchedEnv2048(count, seed, device='cpu', env_base=base, auto_reset=True, lib=OracleLibrary())
        e.reset()
        parts.append(e)
    for _ in range(T):
        whole.step(None)
        for e in parts:
            e.step(None)
    np.testing.assert_array_equal(torch.cat([e.boards for e in parts]).numpy(), whole.boards.numpy())


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
from oracle_backend import OracleLibrary
from simulate_2048_amd.sharding import make_sharded_env, global_stats, rank_info
dist.init_process_group("gloo")
rank, world, _ = rank_info()
env = make_sharded_env(2048 + 4, 123, device="cpu", auto_reset=True, lib=OracleLibrary())
env.reset()
total_reward = 0
episodes = 0
for t in range(40):
    _, r, d = env.step(None)
    total_reward += int(r.sum()); episodes += int(d.sum())
dist.barrier()
stats = global_stats({"reward": total_reward, "episodes": episodes, "envs": env.num_envs})
gathered = [None] * world
dist.all_gather_object(gathered, (env.env_base, env.boards.numpy()))
if rank == 0:
    gathered.sort(key=lambda x: x[0])
    np.save(os.environ["OUT"], np.concatenate([b for _, b in gathered]))
    print("STATS", int(stats["reward"]), int(stats["episodes"]), int(stats["envs"]))
from simulate_2048_amd.sharding import reduce_episode_stats
ep = reduce_episode_stats(env.rollout_episodes(300))
if rank == 0:
    np.save(os.environ["OUT"] + ".stats.npy", ep.numpy())
dist.destroy_process_group()
'''


def test_two_process_gloo_sharding(tmp_path):
    """world_size 2 over gloo: each rank steps its slice, no collective in the loop; the gathered
    boards equal the single-process run and the reduced statistics add up."""
    out = tmp_path / 'boards.npy'
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, REPO=str(REPO), OUT=str(out), OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    total = 2048 + 4
    ref, _ = oracle.reset_batch(total, 123, 0)
    reward = episodes = 0
    for t in range(40):
        res = oracle.step_batch(ref, None, 123, t, flags=3)
        ref = res['boards']
        reward += int(res['reward'].sum())
        episodes += int(res['done'].sum())
    np.testing.assert_array_equal(np.load(out), ref)
    stats = re.search(r'STATS (\d+) (\d+) (\d+)', proc.stdout)
    assert stats and [int(x) for x in stats.groups()] == [reward, episodes, total]
    # per-episode statistics of a further 300-step rollout, reduced over the two ranks == the unsharded oracle run
    # (each rank counts the games in progress from the start of ITS rollout call, like the oracle started on these boards)
    want = oracle.rollout_episodes(ref, 123, 40, 300, flags=3)['stats']
    got = np.load(str(out) + '.stats.npy').view(np.uint64)
    np.testing.assert_array_equal(got, want)


def test_trajectory_buffer_records_in_place():
    """N4: reward/done/legal/obs rows are written by the step call itself; row t+1 describes the board after step t."""
    from simulate_2048_amd.trajectory import TrajectoryBuffer

    n, T, seed = 200, 12, 31
    env = BatchedEnv2048(n, seed, device='cpu', auto_reset=True, lib=OracleLibrary())
    env.reset()
    traj = TrajectoryBuffer(env, T, record_boards=True)
    traj.begin()
    rng = np.random.default_rng(4)
    ref, _ = oracle.reset_batch(n, seed, 0)
    np.testing.assert_array_equal(traj.legal[0].numpy(), oracle.status_batch(ref)['legal'])
    for t in range(T):
        acts = rng.integers(0, 4, size=n).astype(np.uint8)
        traj.actions[t].copy_(torch.from_numpy(acts))
        traj.step(traj.actions[t])  # the row itself: no copy
        res = oracle.step_batch(ref, acts, seed, t, flags=1, want_legal=True, want_obs=True)
        ref = res['boards']
        np.testing.assert_array_equal(traj.reward[t].numpy(), res['reward'])
        np.testing.assert_array_equal(traj.done[t].numpy(), res['done'])
        np.testing.assert_array_equal(traj.legal[t + 1].numpy(), res['legal'])
        np.testing.assert_array_equal(traj.obs[t + 1].numpy(), res['obs'])
        np.testing.assert_array_equal(traj.boards[t + 1].numpy(), ref)
    assert traj.t == T and int(traj.returns().sum()) == int(traj.reward.sum())
    with pytest.raises(IndexError):
        traj.step(None)
    with pytest.raises(ValueError):
        env.step(None, out_reward=torch.zeros(n, dtype=torch.int64))


def test_unseeded_calls_never_share_a_philox_word():
    """fill_cells(state, number_tile=k, seed=None) consumes k consecutive step words, so the entropy
    stream must reserve k indices -- otherwise the first unseeded step after reset() re-used the word of the
    second start tile (its spawn was then fully determined by that tile)."""
    from simulate_2048_amd import runtime
    from simulate_2048_amd.twentyfortyeight.core import _device

    seed_a, first = runtime.entropy_stream(2)
    seed_b, after = runtime.entropy_stream()
    assert seed_a == seed_b and after == first + 2
    assert runtime.entropy_stream(5)[1] == after + 1 and runtime.entropy_stream()[1] == after + 6
    assert _device.rng_for(None, 3)[1] == after + 7 and _device.rng_for(None)[1] == after + 10
    assert _device.rng_for(123, 4) == (123, 0)  # a seeded call always starts its own stream at 0


def test_unseeded_reset_then_step_draw_independent_words():
    """Behavioural form of the same fix, through the mirror on the oracle stand-in: after an unseeded
    fill_cells(zeros, 2) the next unseeded single spawn on a 15-empty board must NOT repeat the (rank, value)
    of the second start tile every time."""
    from simulate_2048_amd import runtime
    from simulate_2048_amd.twentyfortyeight.core import gameboard

    with runtime.use(OracleLibrary(), 'cpu'):
        same = 0
        trials = 300
        for _ in range(trials):
            a = gameboard.fill_cells(np.zeros((4, 4), dtype=np.int64), 2, None)
            first_cells = np.argwhere(a != 0)
            b = np.zeros((4, 4), dtype=np.int64)
            b[tuple(first_cells[0])] = 2  # any board with 15 empties
            before = b.copy()
            gameboard.fill_cells(b, 1, None)
            new_cell = np.argwhere((b != 0) & (before == 0))[0]
            rank = int(((before == 0).reshape(-1)[: 4 * new_cell[0] + new_cell[1]]).sum())
            # the second start tile's rank among the 15 cells left after the first one, and its value
            flat = (a != 0).reshape(-1)
            idx = np.flatnonzero(flat)
            same += int(rank in (idx[1] - 1, idx[0]) and b[tuple(new_cell)] in a)
        assert same < trials * 0.5  # before the fix: 100 %


def test_expectimax_ply_wrapper(golden):
    """ops.expectimax_ply through the stand-in: shapes, optional status, agreement with the separate wrappers."""
    from simulate_2048_amd import ops, runtime

    boards = torch.from_numpy(golden['afterstates']['boards'][:200].copy())
    with runtime.use(OracleLibrary(), 'cpu'):
        ply = ops.expectimax_ply(boards)
        lean = ops.expectimax_ply(boards, want_status=False)
        after, reward, legal = ops.afterstates4(boards)
    assert tuple(ply['boards'].shape) == (200, 4, 32, 16) and tuple(ply['prob'].shape) == (200, 4, 32) and 'status' not in lean
    assert torch.equal(ply['reward'], reward) and torch.equal(ply['legal'], legal) and torch.equal(lean['boards'], ply['boards'])
    np.testing.assert_array_equal(ply['reward'].numpy(), golden['afterstates']['reward'][:200])
    legal_bits = ((legal[:, None] >> torch.arange(4)) & 1).bool()
    assert bool(((ply['count'] > 0) == legal_bits).all())
