"""
Depth-d expectimax (d >= 3) with the search tree kept on the device as one dense frontier per level (DESIGN section 10 "Open (1)"
of round 2; VERDICT r2 next #8 asked for depth d):
    V_d(s) = max over legal a of Q_d(s, a);  Q_d(s, a) = r(s, a) + (0.9 * S2 + 0.1 * S4) / num_empty,  S2 / S4 = sums of V_{d-1}
    over the outcomes that spawned a 2 / a 4.
Parity chain: the REFERENCE's functions composed by plain recursion (tests/golden/expectimax_d3.npz, written by
oracle/gen_golden.py from the imported reference: latent_state, after_state with its probabilities, can_move) => the oracle's C
recursion (float64, ~1e-12: the reference multiplies every outcome by its probability, the build sums first) => the device chain
expand x (d-2) -> depth-2 kernel -> back-up x (d-2), which repeats the oracle's double operations one by one: BIT-EQUAL q / value,
equal best actions.  The host-logic run uses the oracle-backed stand-in of the four entry points (tests/oracle_backend.py).
"""

import numpy as np
import pytest
import torch

import oracle
from simulate_2048_amd import capi, ops, runtime


def mixed_boards(rng, n):
    b = np.where(rng.random((n, 16)) < 0.35, 0, rng.integers(1, 5, size=(n, 16))).astype(np.uint8)
    b[: n // 3] = np.where(rng.random((n // 3, 16)) < 0.1, 0, rng.integers(1, 4, size=(n // 3, 16)))  # crowded: few children, some terminal
    b[0] = 0                                                           # empty board: no legal move at all
    b[1] = np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 1])  # checkerboard: terminal
    b[2] = 0
    b[2, 5] = 3                                                        # a lone tile: every move legal, 15 empty cells each
    b[3] = np.array([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])  # full, ONE merge possible: afterstate with one empty cell
    return b


def test_oracle_depth3_matches_the_reference_composed_fixture(golden):
    f = golden['expectimax_d3']
    res = oracle.expectimax_depth(f['boards'], 3)
    fin = np.isfinite(f['q'])
    assert np.array_equal(np.isfinite(res['q']), fin) and fin.sum() > 100
    np.testing.assert_allclose(res['q'][fin], f['q'][fin], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(res['value'], f['value'], rtol=1e-12, atol=1e-9)
    top = np.where(fin, f['q'], -np.inf).max(axis=1)
    for i in np.flatnonzero(res['best'] != f['best']):  # only on ties up to rounding: from depth 3 on both sides sum floats,
        assert abs(f['q'][i, res['best'][i]] - top[i]) <= 1e-9  # so which of two mathematically equal actions wins is noise
    assert (res['best'] != f['best']).sum() <= 4
    # depth 2 through the same entry point == the depth-2 call
    d2 = oracle.expectimax_depth2(f['boards'])
    r2 = oracle.expectimax_depth(f['boards'], 2)
    assert np.array_equal(r2['q'], d2['q']) and np.array_equal(r2['best'], d2['best']) and np.array_equal(r2['value'], d2['value'])


def test_oracle_depth4_matches_the_reference_composed_fixture(golden):
    """One level deeper (VERDICT r3 weak #2d: depth 4 was checked against the oracle's recursion alone): the reference's functions
    composed by plain recursion to depth 4 on 24 dense boards (tests/golden/expectimax_d4.npz)."""
    f = golden['expectimax_d4']
    res = oracle.expectimax_depth(f['boards'], 4)
    fin = np.isfinite(f['q'])
    assert np.array_equal(np.isfinite(res['q']), fin) and fin.sum() >= 24
    np.testing.assert_allclose(res['q'][fin], f['q'][fin], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(res['value'], f['value'], rtol=1e-12, atol=1e-9)
    top = np.where(fin, f['q'], -np.inf).max(axis=1)
    for i in np.flatnonzero(res['best'] != f['best']):  # ties up to rounding only
        assert abs(f['q'][i, res['best'][i]] - top[i]) <= 1e-9


def test_expectimax_depth4_on_the_reference_fixture(api_runtime, golden):
    f = golden['expectimax_d4']
    out = ops.expectimax(torch.from_numpy(f['boards']).to(api_runtime.device), 4)
    fin = np.isfinite(f['q'])
    np.testing.assert_allclose(out['q'].cpu().numpy()[fin], f['q'][fin], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(out['value'].cpu().numpy(), f['value'], rtol=1e-12, atol=1e-9)
    ref = oracle.expectimax_depth(f['boards'], 4)
    assert np.array_equal(out['q'].cpu().numpy(), ref['q']) and np.array_equal(out['best'].cpu().numpy(), ref['best'])  # bit for bit


def test_expand_builds_the_dense_frontier(api_runtime):
    """ops.expand: children in (action, empty cell rank, tile 2 then 4) order, offsets, rewards, empty counts -- against the
    oracle's afterstates and chance outcomes (after_state's order, gameboard.py:132-171)."""
    rng = np.random.default_rng(5)
    b = mixed_boards(rng, 300)
    out = ops.expand(torch.from_numpy(b).to(api_runtime.device))
    kids, off = out['children'].cpu().numpy(), out['offsets'].cpu().numpy()
    reward, ne = out['reward'].cpu().numpy(), out['n_empty'].cpu().numpy()
    at = 0
    for i in range(b.shape[0]):
        assert off[i] == at
        for a in range(4):
            res = oracle.afterstate_batch(b[i:i + 1], np.array([a], dtype=np.uint8))
            if not res['changed'][0]:
                assert ne[i, a] == 0 and reward[i, a] == 0
                continue
            after = res['boards'][0]
            cells = np.flatnonzero(after == 0)
            assert ne[i, a] == len(cells) >= 1 and reward[i, a] == res['reward'][0]
            for cell in cells:
                for e in (1, 2):
                    want = after.copy()
                    want[cell] = e
                    assert np.array_equal(kids[at], want), (i, a, cell, e)
                    at += 1
    assert at == kids.shape[0]
    assert ne[0].sum() == 0 and ne[1].sum() == 0 and (ne[2] == 15).all() and sorted(ne[3].tolist()) == [0, 0, 1, 1]
    empty = ops.expand(torch.zeros((0, 16), dtype=torch.uint8, device=api_runtime.device))
    assert empty['children'].shape == (0, 16) and empty['offsets'].numel() == 0


@pytest.mark.parametrize('depth,n', [(2, 60), (3, 40), (4, 6)])
def test_expectimax_equals_the_oracle_recursion_bit_for_bit(api_runtime, depth, n):
    rng = np.random.default_rng(100 + depth)
    b = mixed_boards(rng, max(n, 6))[:max(n, 6)]
    out = ops.expectimax(torch.from_numpy(b).to(api_runtime.device), depth)
    ref = oracle.expectimax_depth(b, depth)
    assert np.array_equal(out['q'].cpu().numpy(), ref['q'])  # float64, every operation repeated in the same order
    assert np.array_equal(out['best'].cpu().numpy(), ref['best'])
    assert np.array_equal(out['value'].cpu().numpy(), ref['value'])
    assert out['best'][0] == 255 and out['best'][1] == 255 and float(out['value'][0]) == 0.0
    assert len(out['frontier']) == depth - 1 and out['frontier'][0] == b.shape[0]
    if depth >= 3:
        assert out['frontier'][1] > 10 * b.shape[0] // 2  # the frontier really grows


def test_expectimax_on_the_reference_fixture(api_runtime, golden):
    f = golden['expectimax_d3']
    out = ops.expectimax(torch.from_numpy(f['boards']).to(api_runtime.device), 3)
    fin = np.isfinite(f['q'])
    np.testing.assert_allclose(out['q'].cpu().numpy()[fin], f['q'][fin], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(out['value'].cpu().numpy(), f['value'], rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        ops.expectimax(torch.from_numpy(f['boards']).to(api_runtime.device), 1)


@pytest.mark.gpu
def test_depth3_at_scale_matches_the_oracle():
    """2000 mid-game roots at depth 3 (about 70 000 leaves valued by the depth-2 kernel) against the OpenMP oracle, bit for bit;
    argument checks of the new entry points."""
    lib = capi.library()
    rt = runtime.Runtime(lib, 'cuda:0')
    n = 2000
    boards = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.reset(0, s, n, 0, boards.data_ptr(), None, 77, 0)
    lib.rollout(0, s, n, 0, boards.data_ptr(), 77, 0, 120, None, None, None, 3)
    out = ops.expectimax(boards, 3, rt=rt)
    ref = oracle.expectimax_depth(boards.cpu().numpy(), 3)
    assert np.array_equal(out['q'].cpu().numpy(), ref['q']) and np.array_equal(out['best'].cpu().numpy(), ref['best'])
    assert np.array_equal(out['value'].cpu().numpy(), ref['value'])
    assert out['frontier'][1] > 20 * n
    v = torch.zeros((n,), dtype=torch.float64, device='cuda')
    lib.expectimax_depth2_f64(0, s, n, boards.data_ptr(), v.data_ptr())
    d2 = oracle.expectimax_depth2(boards.cpu().numpy())
    assert np.array_equal(v.cpu().numpy(), d2['value'])
    with pytest.raises(capi.Env2048Error) as e:
        lib.expand_count(0, s, n, boards.data_ptr(), None)
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.Env2048Error) as e:
        lib.expectimax_depth2_f64(0, s, n, boards.data_ptr(), v.data_ptr() + 4)
    assert e.value.code == capi.ERR_ALIGNMENT


def test_all_terminal_roots_have_empty_frontiers(api_runtime):
    """Roots without a legal move: every frontier is empty, nothing is launched on it, values are 0 and best is 255."""
    b = np.tile(np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 1], dtype=np.uint8), (5, 1))
    out = ops.expectimax(torch.from_numpy(b).to(api_runtime.device), 4)
    assert out['frontier'] == [5, 0, 0]
    assert (out['best'].cpu().numpy() == 255).all() and not out['value'].cpu().numpy().any()
    assert np.isneginf(out['q'].cpu().numpy()).all()


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 511, 512, 513, 70_001, 700_003])
def test_expand_offsets_on_the_device_equal_count_and_cumsum(n):
    """env2048_expand_offsets (block scan, scan of the block totals by ONE workgroup -- 700 003 boards = 1368 blocks, more than
    its 1024 threads --, add) against env2048_expand_count + a host prefix sum; the total arrives in pinned host memory behind
    its completion word, and in scratch[blocks]."""
    lib = capi.library()
    s = torch.cuda.current_stream().cuda_stream
    boards = torch.zeros((n, 16), dtype=torch.uint8, device='cuda')
    lib.reset(0, s, n, 5, boards.data_ptr(), None, 31, 0)
    lib.rollout(0, s, n, 5, boards.data_ptr(), 31, 0, 60 + n % 90, None, None, None, 3)
    counts = torch.zeros((n,), dtype=torch.int32, device='cuda')
    lib.expand_count(0, s, n, boards.data_ptr(), counts.data_ptr())
    want = np.cumsum(counts.cpu().numpy().astype(np.int64))
    offsets = torch.full((n,), -1, dtype=torch.int64, device='cuda')
    blocks = (n + 511) // 512
    scratch = torch.full((blocks + 1,), -1, dtype=torch.int64, device='cuda')
    slot = torch.zeros((2,), dtype=torch.int64).pin_memory()
    lib.expand_offsets(0, s, n, boards.data_ptr(), offsets.data_ptr(), scratch.data_ptr(), slot.data_ptr(), 4242)
    assert lib.wait_sequence(slot.data_ptr() + 8, 4242, 2_000_000)
    assert int(slot[0]) == int(want[-1])  # visible as soon as the word is (no synchronisation in between)
    torch.cuda.synchronize()
    assert np.array_equal(offsets.cpu().numpy(), want - counts.cpu().numpy())
    assert int(scratch[blocks]) == int(want[-1])
    lib.expand_offsets(0, s, n, boards.data_ptr(), offsets.data_ptr(), scratch.data_ptr(), None, 0)  # without the host slot
    torch.cuda.synchronize()
    assert np.array_equal(offsets.cpu().numpy(), want - counts.cpu().numpy())
    with pytest.raises(capi.Env2048Error):
        lib.expand_offsets(0, s, n, boards.data_ptr(), offsets.data_ptr(), None, None, 0)
