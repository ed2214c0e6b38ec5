"""
Multi-GPU layout of a batch: the global environment index range
[0, total) is cut into contiguous slices, one per rank; every rank steps its own slice with the
same (seed, step_index) and ``env_base`` = first global index of the slice.  Because the Philox
counter carries the GLOBAL index, an N-environment job produces byte-identical boards for any
number of ranks -- there is no collective on the hot path (no RCCL, no xGMI traffic).

``torch.distributed`` appears only for optional end-of-run statistics and for the benchmark's
barrier / max-over-ranks timing.
"""

from __future__ import annotations

import os

import torch

from .batch import BatchedEnv2048


def shard_range(total: int, rank: int, world_size: int, multiple: int = 4) -> tuple[int, int]:
    """(env_base, count) of rank's slice.  Slice boundaries are rounded to `multiple` environments
    (4 = one Philox block) so no rank pays for a straddling block; the last rank takes the tail."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} not in [0, {world_size})')
    per = -(-total // world_size)
    per = -(-per // multiple) * multiple
    lo = min(rank * per, total)
    hi = min(lo + per, total)
    return lo, hi - lo


def rank_info() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process when absent)."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def make_sharded_env(total_envs: int, seed: int, *, rank: int | None = None, world_size: int | None = None,
                     device=None, **kwargs) -> BatchedEnv2048:
    """This rank's slice of a `total_envs` job as a BatchedEnv2048 (one process per GPU)."""
    env_rank, env_world, local_rank = rank_info()
    rank = env_rank if rank is None else rank
    world_size = env_world if world_size is None else world_size
    env_base, count = shard_range(total_envs, rank, world_size)
    if device is None:
        device = torch.device('cuda', local_rank)
    return BatchedEnv2048(count, seed, device=device, env_base=env_base, **kwargs)


def global_stats(local: dict, group=None) -> dict:
    """Sum per-rank scalar statistics (e.g. total reward, episode count) over all ranks: a handful of
    scalars through one all_reduce -- off the hot path."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return {k: float(v) for k, v in local.items()}
    keys = sorted(local)
    backend = dist.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return dict(zip(keys, t.tolist()))


def reduce_episode_stats(stats: torch.Tensor, group=None) -> torch.Tensor:
    """Combine the 40-word episode-statistics blocks of ``BatchedEnv2048.rollout_episodes`` over all ranks (sums, except
    word 4 = minimum and word 5 = maximum of the total reward): three tiny all_reduces at the END of an evaluation run,
    off the hot path.  Returns a new int64 tensor on the device of ``stats`` (unchanged without a process group)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return stats.clone()
    backend = dist.get_backend(group)
    work = stats.to(torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else 'cpu').clone()
    # the words are uint64 bit patterns in an int64 tensor: sums wrap identically; min/max need the unsigned order, which
    # for values below 2^63 (any real reward) is the signed order -- except the "no game yet" minimum UINT64_MAX = -1
    lo = work[4:5].clone()
    lo[lo < 0] = torch.iinfo(torch.int64).max
    hi = work[5:6].clone()
    dist.all_reduce(work, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    work[4] = -1 if int(lo) == torch.iinfo(torch.int64).max else lo[0]
    work[5] = hi[0]
    return work.to(stats.device)
