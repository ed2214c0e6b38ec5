"""
``ShardedEnv2048``: one process, G contiguous slices of one global batch, one worker thread + stream per shard, no collective
(VERDICT r2 missing #3).  The sharded run must equal the unsharded batch byte for byte -- on the oracle-backed
stand-in with three "devices" (host logic, CPU) and on the GPU with two shards on device 0.
Partitioning rule restated from SURVEY 8(e): slice g = global indices [g * ceil(N / G), ...), Philox counters carry the GLOBAL index.
"""

import threading

import numpy as np
import pytest
import torch

from oracle_backend import OracleLibrary
from simulate_2048_amd import BatchedEnv2048
from simulate_2048_amd.sharded import ShardedEnv2048


@pytest.fixture(params=['host-logic', pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hip':
        return dict(devices=['cuda:0', 'cuda:0'], lib=None, one='cuda:0')
    return dict(devices=['cpu', 'cpu', 'cpu'], lib=OracleLibrary(), one='cpu')


def test_sharded_equals_unsharded(backend):
    n, seed, base = 10_007, 13, 3  # odd size: the last shard is shorter; odd base: shard boundaries straddle Philox blocks
    whole = BatchedEnv2048(n, seed, device=backend['one'], env_base=base, auto_reset=True, with_legal=True, lib=backend['lib'])
    sharded = ShardedEnv2048(n, seed, devices=backend['devices'], env_base=base, auto_reset=True, with_legal=True, lib=backend['lib'])
    try:
        assert sharded.num_shards == len(backend['devices']) and sum(c for _, c in sharded.ranges) == n
        assert [e.env_base for e in sharded.shards] == [base + lo for lo, _ in sharded.ranges]
        whole.reset()
        sharded.reset()
        assert torch.equal(sharded.gather('boards'), whole.boards.cpu())
        rng = np.random.default_rng(2)
        for t in range(6):
            if t % 2:
                acts = torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8))
                whole.step(acts.to(whole.device))
                sharded.step(sharded.split(acts))  # per-shard tensors on their devices
            else:
                whole.step(None)
                sharded.step(None)
            for name in ('boards', 'reward', 'done', 'legal'):
                assert torch.equal(sharded.gather(name), getattr(whole, name).cpu()), (t, name)
        assert sharded.step_index == whole.step_index == 6
        # K dependent steps enqueued per shard by one C call from its own thread
        sharded.step_sequence(5)
        for _ in range(5):
            whole.step(None)
        assert torch.equal(sharded.gather('boards'), whole.boards.cpu()) and sharded.step_index == whole.step_index
        staged = torch.from_numpy(rng.integers(0, 4, size=(4, n)).astype(np.uint8))
        sharded.step_sequence(4, [staged[:, lo:lo + c].contiguous().to(d) for (lo, c), d in zip(sharded.ranges, sharded.devices)])
        for k in range(4):
            whole.step(staged[k].to(whole.device))
        assert torch.equal(sharded.gather('boards'), whole.boards.cpu()) and torch.equal(sharded.gather('reward'), whole.reward.cpu())
        # fused rollout + statistics reduced on the host
        a = whole.rollout(40)
        b = sharded.rollout(40)
        assert torch.equal(torch.cat([x.cpu() for x in b['return_sum']]), a['return_sum'].cpu())
        assert torch.equal(sharded.gather('boards'), whole.boards.cpu())
        st_whole = BatchedEnv2048.episode_statistics(whole.rollout_episodes(300))
        st_sharded = sharded.episode_statistics(sharded.rollout_episodes(300))
        assert st_sharded == st_whole and st_whole['games'] > 0
        ms = sharded.timed(lambda g: sharded.shards[g].step(None))
        assert len(ms) == sharded.num_shards and all(m >= 0 for m in ms)
        with pytest.raises(ValueError):
            sharded.step([torch.zeros(3, dtype=torch.uint8)])
    finally:
        sharded.close()


def test_worker_errors_reach_the_caller(backend):
    sharded = ShardedEnv2048(64, 1, devices=backend['devices'], lib=backend['lib'])
    try:
        with pytest.raises(ValueError, match='actions'):
            sharded.step([torch.zeros(5, dtype=torch.uint8, device=d) for d in sharded.devices])  # wrong per-shard length
        sharded.reset()  # the workers are still alive afterwards
        sharded.step(None)
    finally:
        sharded.close()
    with pytest.raises(ValueError):
        ShardedEnv2048(64, 1, devices=backend['devices'], lib=backend['lib'], device_step_counter=True)


def test_eight_shards_each_on_its_own_device_and_stream(monkeypatch):
    """G = 8 (fake) devices and an odd total -- every C-ABI call that touches shard g's slice is made with
    device ordinal g and shard g's stream, never another's.  There is no 8-GPU box here, so the devices are faked where the
    product reads them: ``batch._stream_handle`` (device ordinal + current stream of the CALLING thread) answers for the worker
    thread it is called on, which is exactly what makes the real thing correct -- worker g has device g and its stream current,
    and shard g's methods run only there.  The oracle-backed stand-in records (device, stream, env_base, n) of every call."""
    from simulate_2048_amd import batch
    from simulate_2048_amd import sharded as sharded_module

    G, n, seed, base = 8, 100_003, 21, 5
    lib = OracleLibrary()

    def fake_handle(device):
        name = threading.current_thread().name
        assert name.startswith('env2048-shard-'), f'a C-ABI call for a shard was made from thread {name!r}, not from a shard worker'
        g = int(name.rsplit('-', 1)[1])
        return g, 0xA000 + g

    monkeypatch.setattr(batch, '_stream_handle', fake_handle)
    monkeypatch.setattr(sharded_module, '_stream_handle', fake_handle)  # (imported by name there)
    sharded = ShardedEnv2048(n, seed, devices=['cpu'] * G, env_base=base, auto_reset=True, lib=lib)
    try:
        assert sharded.num_shards == G and sum(c for _, c in sharded.ranges) == n
        assert all(lo % 4 == 0 for lo, _ in sharded.ranges) and sharded.ranges[-1][1] < sharded.ranges[0][1]  # Philox-block aligned, short tail
        sharded.reset()
        sharded.step(None)
        rng = np.random.default_rng(1)
        acts = torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8))
        sharded.step(sharded.split(acts))
        sharded.step_sequence(3)
        sharded.rollout(5)
        whole = BatchedEnv2048(n, seed, device='cpu', env_base=base, auto_reset=True, lib=OracleLibrary())
        monkeypatch.setattr(batch, '_stream_handle', lambda device: (-1, None))
        whole.reset()
        whole.step(None)
        whole.step(acts)
        for _ in range(3):
            whole.step(None)
        whole.rollout(5)
        assert torch.equal(sharded.gather('boards'), whole.boards)  # and the 8 slices are the unsharded batch, byte for byte
    finally:
        sharded.close()
    by_shard = {base + lo: (g, c) for g, (lo, c) in enumerate(sharded.ranges)}
    seen = set()
    assert len(lib.launches) >= G * 7
    for name, device, stream, env_base, count in lib.launches:
        g, c = by_shard[env_base]  # every call starts at a shard boundary ...
        assert count == c and device == g and stream == 0xA000 + g, (name, device, stream, env_base, count)  # ... and is that shard's own
        seen.add((name, g))
    assert {g for _, g in seen} == set(range(G)) and {name for name, _ in seen} >= {'reset', 'step', 'rollout'}


def test_checkpoint_and_resume_continue_the_same_random_stream(backend, tmp_path):
    """state_dict() / load_state_dict() (checkpoint / resume: the env state is boards + (seed, step index,
    first global index) + the running episode of the statistics): a batch restored from a ``torch.save``d checkpoint, into a
    fresh object, produces byte for byte what the original produces -- single batch and sharded, statistics carry included."""
    n, seed, base = 5003, 99, 17
    a = BatchedEnv2048(n, seed, device=backend['one'], env_base=base, auto_reset=True, with_legal=True, lib=backend['lib'])
    a.reset()
    for _ in range(7):
        a.step(None)
    stats_a = a.rollout_episodes(40)
    path = tmp_path / 'env.pt'
    torch.save(a.state_dict(), path)
    b = BatchedEnv2048(n, 1, device=backend['one'], lib=backend['lib'], with_legal=True)  # other seed / base / policy: all restored
    b.load_state_dict(torch.load(path, weights_only=True))
    assert (b.seed, b.env_base, b.step_index, b.auto_reset) == (a.seed, a.env_base, a.step_index, True)
    assert torch.equal(b.boards, a.boards) and torch.equal(b.legal, a.status()['legal'])  # (legal / obs are re-derived from the restored boards)
    stats_b = stats_a.clone()
    a.rollout_episodes(60, stats_a)
    b.rollout_episodes(60, stats_b)  # the games in progress at the checkpoint are counted whole on both sides
    assert torch.equal(stats_a, stats_b) and torch.equal(a.boards, b.boards)
    for _ in range(5):
        a.step(None)
        b.step(None)
    assert torch.equal(a.boards, b.boards) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    with pytest.raises(ValueError):
        BatchedEnv2048(n + 1, 1, device=backend['one'], lib=backend['lib']).load_state_dict(a.state_dict())
    sh = ShardedEnv2048(n, seed, devices=backend['devices'], env_base=base, auto_reset=True, lib=backend['lib'])
    sh2 = ShardedEnv2048(n, 5, devices=backend['devices'], lib=backend['lib'])
    try:
        sh.reset()
        sh.step_sequence(9)
        sh2.load_state_dict(sh.state_dict())
        sh.step_sequence(6)
        sh2.step_sequence(6)
        assert torch.equal(sh.gather('boards'), sh2.gather('boards')) and sh2.step_index == sh.step_index
        with pytest.raises(ValueError):
            ShardedEnv2048(n + 4, 5, devices=backend['devices'], lib=backend['lib']).load_state_dict(sh.state_dict())
    finally:
        sh.close()
        sh2.close()
