#!/usr/bin/env python3
"""Host + device cost of ONE MOVE of the reference's self-play loop on the ``reinforce.game`` mirrors (GPU box).

A move, as reinforce/training/self_play.py:316-317,346-356 asks the environment for it:
    observations = batched_get_observation(states); legal_masks = batched_get_legal_actions(states)
    np.array(observations)                       # the caller's own device -> host copies are NOT env work and are timed apart
    states, rewards, dones, _ = batched_step(states, actions, step_keys)   # actions / keys arrive as host (N,) / (N, 2) arrays
``env_work_us`` = wall time of the three environment calls per move with the stream kept busy (no per-move synchronisation: what
the env costs the loop's host thread), measured over `moves` moves; ``move_with_readback_us`` adds the caller's np.array(...)
conversions (one synchronisation per move, as in the reference).  Keys and actions are NumPy arrays (the worst case for the
mirrors: they have to reach the device), or device tensors (``dev_inputs``)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import resultpool  # noqa: E402
from simulate_2048_amd.reinforce.game import env as genv  # noqa: E402


def run(n: int, moves: int, dev_inputs: bool, readback: bool) -> float:
    rng = np.random.default_rng(n)
    keys = [rng.integers(0, 1 << 32, size=(n, 2), dtype=np.uint64).astype(np.uint32) for _ in range(64)]
    acts = [rng.integers(0, 4, size=n).astype(np.int32) for _ in range(64)]
    if dev_inputs:
        keys = [torch.from_numpy(k.view(np.int32)).cuda() for k in keys]
        acts = [torch.from_numpy(a).cuda() for a in acts]
    states = genv.batched_reset(keys[0])
    for warm in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for m in range(moves):
            observations = genv.batched_get_observation(states)
            legal_masks = genv.batched_get_legal_actions(states)
            if readback:
                np.array(observations), np.array(legal_masks)
            states, rewards, dones, _ = genv.batched_step(states, acts[m & 63], keys[m & 63])
            if readback:
                np.array(rewards), np.array(dones)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / moves * 1e6
    return dt


def main():
    out = {'note': __doc__.split('\n')[0]}
    for n in (2, 8, 64, 4096, 1 << 16):
        moves = 2000 if n <= 4096 else 500
        out[f'N={n}'] = {
            'env_work_us_numpy_inputs': round(run(n, moves, False, False), 2),
            'env_work_us_device_inputs': round(run(n, moves, True, False), 2),
            'move_with_readback_us_numpy_inputs': round(run(n, moves, False, True), 2),
        }
    out['readback'] = ('np.array(...) of a pooled result reads the set\'s pinned host mirror, which the launch itself wrote '
                       '(env2048_step_state_i32_keyed_signalled): the first read after a launch polls the launch\'s completion word in the '
                       f'mirror (n <= 512; falls back to one stream synchronisation after {resultpool.SIGNAL_TIMEOUT_US} us or for larger n), '
                       'further reads of that state are host copies (resultpool.ResultSet.host_array; sets above '
                       f'{resultpool.MIRROR_MAX_BYTES >> 10} KiB are read array by array, one device-to-host copy each)')
    out['result_sets_created'] = resultpool.pool.created
    sets = [rs for group in resultpool.pool._sets.values() for rs in group]
    out['first_reads_served_by'] = {'completion_word': sum(rs.signal_hits for rs in sets), 'stream_synchronize': sum(rs.stream_syncs for rs in sets),
                                    'device_to_host_copy': sum(rs.mirror_copies for rs in sets)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
