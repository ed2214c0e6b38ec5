"""
Host logic of the C-ABI that needs no GPU: the launch planner of env2048_step_many (csrc/step_planner.hpp) and the helper
thread of env2048_step_sequence (csrc/helper_thread.hpp).

* regression for the round-3 hang: a window of 64 empty records followed by more made `for_each_launch` loop forever
  (ADVICE r3); the plan call now runs in a CHILD process under a 2-second watchdog;
* a hypothesis property test of env2048_step_many_plan through the built library (the plan touches no device): it
  terminates, launch ids are non-decreasing, at most 32 live records per launch, no two records of a launch touch
  overlapping memory with a write (brute force restated HERE from the header's contract, not from the planner's code),
  empty records anywhere;
* the same headers compiled by g++ with ASan + UBSan (tests/host_emu/planner_main.cpp): a C++ fuzz of the planner and the
  helper thread's ordering / fork / poll-window behaviour with stub jobs.
"""

import ctypes
import subprocess
import sys

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import REPO
from simulate_2048_amd import capi

B, A, R, D, L, O, M, C = (0x1000_0000_0000 * k for k in range(1, 9))
MAX_FUSED = 32


def rec(n, slot=0, *, stride=4096, flags=capi.FLAG_AUTO_RESET, legal=False, obs=False, max_exp=False, counter=False, reward=True,
        action_slot=None, done_slot=None, boards_in_rewards=False):
    a_slot = slot if action_slot is None else action_slot
    d_slot = slot if done_slot is None else done_slot
    boards = (R + slot * stride * 16) if boards_in_rewards else (B + slot * stride * 16)  # (inside the reward arena: aliases other records' outputs)
    return capi.StepArgs(n, 0, boards if n else 0, (A + a_slot * stride) if n else 0, (R + slot * stride * 4) if (n and reward) else 0,
                         (D + d_slot * stride) if n else 0, (L + slot * stride) if (n and legal) else 0,
                         (O + slot * stride * 64) if (n and obs) else 0, 7, 0, flags, 0, (M + slot * stride) if (n and max_exp) else 0,
                         C if (n and counter) else 0)


def plan(records):
    arr = (capi.StepArgs * len(records))(*records)
    return capi.library().step_many_plan(arr)


def ranges_of(a):
    """(lo, hi, write) byte ranges of one record, from the contract in include/env2048.h (env2048_step_many)."""
    out = []
    if a.n == 0:
        return out
    out.append((a.boards, a.boards + 16 * a.n, True))
    if not a.flags & capi.FLAG_RANDOM_ACTIONS and a.actions:
        out.append((a.actions, a.actions + a.n, False))
    if a.reward:
        out.append((a.reward, a.reward + 4 * a.n, True))
    out.append((a.done, a.done + a.n, True))
    if a.legal:
        out.append((a.legal, a.legal + a.n, True))
    if a.obs:
        out.append((a.obs, a.obs + a.n * 16 * (2 if a.flags & (capi.FLAG_OBS_F16 | capi.FLAG_OBS_BF16) else 4), True))
    if a.max_exp:
        out.append((a.max_exp, a.max_exp + a.n, True))
    return out


def dependent(x, y):
    return any((wx or wy) and lx < hy and ly < hx for lx, hx, wx in ranges_of(x) for ly, hy, wy in ranges_of(y))


def check_plan(records, p):
    count = len(records)
    launch_of = p['launch_of']
    assert len(launch_of) == count and launch_of[0] == 0
    assert all(0 <= launch_of[k] - launch_of[k - 1] <= 1 for k in range(1, count))
    live_ids = set()
    for lid in sorted(set(launch_of)):
        members = [records[k] for k in range(count) if launch_of[k] == lid and records[k].n]
        assert len(members) <= MAX_FUSED
        if members:
            live_ids.add(lid)
        if len(members) > 1:
            assert not any(m.step_counter for m in members)
            assert len({bool(m.legal or m.obs or m.max_exp or m.flags & capi.FLAG_REWARD_LOG2_F32) for m in members}) == 1
            assert len({m.flags & capi.FLAG_RANDOM_ACTIONS for m in members}) == 1
            assert len({m.flags & (capi.FLAG_TILE_SMALL | capi.FLAG_TILE_LARGE) for m in members}) == 1
        for i, x in enumerate(members):
            for y in members[i + 1:]:
                assert not dependent(x, y), (lid, x.boards, y.boards)
    assert p['launches'] == len(live_ids)
    n_live = sum(1 for r in records if r.n)
    assert (p['launches'] > 0) == (n_live > 0) and p['launches'] <= n_live
    assert all(b in (1, 4) for b in p['boards_per_lane'])


# ---- regression: the round-3 hang ------------------------------------------------------------------------------------------
HANG_CASES = {
    '64_empty_then_6_live': [0] * 64 + [1] * 6,
    '32_empty_then_40_live': [0] * 32 + [1] * 40,
    '64_empty_then_1_live': [0] * 64 + [1],
    '200_empty': [0] * 200,
    'live_64_empty_live': [1] + [0] * 64 + [1] * 5,
    'alternating_300': [k % 2 for k in range(300)],
    '130_empty_between_runs': [1] * 40 + [0] * 130 + [1] * 40,
}

CHILD = r'''
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from test_planner import HANG_CASES, rec, plan, check_plan
for name, shape in HANG_CASES.items():
    records = [rec(4096 * live, slot=k) for k, live in enumerate(shape)]
    p = plan(records)
    check_plan(records, p)
    print(name, p['launches'])
'''


def test_step_many_plan_terminates_on_long_runs_of_empty_records():
    """Each shape hung the round-3 planner.  Run in a child process so that a hang is a 2-second failure, not a stuck suite."""
    proc = subprocess.run([sys.executable, '-c', CHILD % (str(REPO), str(REPO / 'tests'))], capture_output=True, text=True, timeout=30)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    got = dict(line.split() for line in proc.stdout.splitlines())
    # independent 4096-board records on slots of their own: 32 per launch; empty records launch nothing and separate nothing
    assert got == {'64_empty_then_6_live': '1', '32_empty_then_40_live': '2', '64_empty_then_1_live': '1', '200_empty': '0',
                   'live_64_empty_live': '1', 'alternating_300': '5', '130_empty_between_runs': '3'}


def test_the_watchdog_itself_would_catch_a_hang():
    with pytest.raises(subprocess.TimeoutExpired):
        subprocess.run([sys.executable, '-c', 'while True: pass'], timeout=2)


def test_empty_records_take_the_launch_of_the_run_they_sit_in():
    records = [rec(0), rec(0), rec(4096, 0), rec(0), rec(4096, 1), rec(4096, 0), rec(0), rec(0)]
    p = plan(records)
    assert p['launch_of'] == [0, 0, 0, 0, 0, 1, 1, 1] and p['launches'] == 2
    assert plan([rec(0), rec(0)]) == {'launches': 0, 'launch_of': [0, 0], 'boards_per_lane': [1, 1], 'write_through': [True, True]}
    # an empty record is validated like any other (flags, reserved) although it launches nothing
    bad = rec(0)
    bad.flags = 1 << 20
    with pytest.raises(capi.Env2048Error, match='unknown flag bits'):
        plan([rec(4096), bad])


# ---- property test through the built library --------------------------------------------------------------------------------
record_strategy = st.builds(
    rec,
    n=st.one_of(st.just(0), st.integers(1, 4096), st.sampled_from([1, 64, 2048, 4096])),
    slot=st.integers(0, 11),
    stride=st.sampled_from([4096, 2049, 1000]),  # (smaller than n: neighbouring slots overlap)
    flags=st.sampled_from([0, capi.FLAG_AUTO_RESET, capi.FLAG_AUTO_RESET | capi.FLAG_RANDOM_ACTIONS, capi.FLAG_TILE_SMALL,
                           capi.FLAG_AUTO_RESET | capi.FLAG_TILE_LARGE, capi.FLAG_OBS_BF16]),
    legal=st.booleans(), obs=st.sampled_from([False, False, True]), max_exp=st.sampled_from([False, False, False, True]),
    counter=st.sampled_from([False] * 9 + [True]), reward=st.sampled_from([True, True, False]),
    action_slot=st.one_of(st.none(), st.integers(0, 11)), done_slot=st.one_of(st.none(), st.integers(0, 11)),
    boards_in_rewards=st.sampled_from([False] * 7 + [True]),
)


def empty_runs(records, data):
    """Splice runs of 1..140 empty records into a record list (the shape no round-3 fuzzer drew)."""
    out = list(records)
    for _ in range(data.draw(st.integers(0, 3))):
        at = data.draw(st.integers(0, len(out)))
        out[at:at] = [rec(0)] * data.draw(st.integers(1, 140))
    return out


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(st.lists(record_strategy, min_size=1, max_size=80), st.data())
def test_step_many_plan_properties(records, data):
    records = empty_runs(records, data)
    check_plan(records, plan(records))


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(1, 150), st.integers(1, 4096), st.data())
def test_long_independent_runs_are_cut_every_32_live_records(count, n, data):
    records = empty_runs([rec(n, slot=k) for k in range(count)], data)
    p = plan(records)
    check_plan(records, p)
    assert p['launches'] == (count + MAX_FUSED - 1) // MAX_FUSED


# ---- the same headers under ASan + UBSan ------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def planner_main():
    src = REPO / 'tests' / 'host_emu' / 'planner_main.cpp'
    out = REPO / 'tests' / 'host_emu' / '_build' / 'planner_main'
    csrc = REPO / 'simulate_2048_amd' / 'csrc'
    deps = [src, csrc / 'step_planner.hpp', csrc / 'helper_thread.hpp', csrc / 'errors.hpp', csrc / 'tiling.hpp', REPO / 'include' / 'env2048.h']
    out.parent.mkdir(exist_ok=True)
    if not out.exists() or out.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Wextra',
                        '-pthread', f'-I{csrc}', f'-I{REPO / "include"}', str(src), '-o', str(out)], check=True)
    return out


@pytest.mark.parametrize('seed', [1, 20261005])
def test_planner_fuzz_under_sanitizers(planner_main, seed):
    proc = subprocess.run([str(planner_main), 'plan', str(seed), '2500'], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert 'interval set ok' in proc.stdout and 'largest launch 32 records' in proc.stdout


def test_helper_thread_order_fork_and_poll_window(planner_main):
    """Stub jobs (no HIP): results in order, two concurrent callers served one at a time, a forked child builds its own
    helper, a 0-microsecond poll window burns no CPU while idle."""
    proc = subprocess.run([str(planner_main), 'helper'], capture_output=True, text=True, timeout=60)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert 'helper thread ok' in proc.stdout


def test_helper_stats_entry_point_needs_no_device():
    info = capi.library().helper_stats(0)
    assert info['created'] is False and info['jobs'] == 0 and info['cpu_ns'] == 0 and 0 <= info['spin_us'] <= 100000
    out_of_range = capi.library().helper_stats(99)
    assert out_of_range['created'] is False
