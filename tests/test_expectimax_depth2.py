"""
Depth-2 expectimax of the true environment on device (VERDICT r2 "next" #8; SURVEY 8(f) N3 widened):
    Q2(s, a) = r(s, a) + sum_o P(o) * max_{legal a'} r(o, a'),   V2 = max over legal a,   best = smallest maximising action.

Parity chain: the REFERENCE's own functions composed literally (latent_state, after_state, can_move: gameboard.py:105-171,
gamemove.py:132-164; fixture tests/golden/expectimax_d2.npz written by oracle/gen_golden.py from the imported reference)
=> the oracle's C restatement (exact integer terms; float64 Q within summation-order noise) => the HIP kernel through the
C-ABI (integer terms exact; Q is evaluated in double from those integers on both sides, so float32 q and best are EQUAL).
"""

import numpy as np
import pytest
import torch

import oracle
from simulate_2048_amd import capi, ops, runtime

DEV = 0


def check_against_reference_fixture(res, f):
    """`res`: terms / q / best / value arrays of an implementation; `f`: the reference-composed fixture."""
    np.testing.assert_array_equal(res['terms'], f['terms'])  # exact integers
    fin = np.isfinite(f['q'])
    assert np.array_equal(np.isfinite(res['q']), fin)  # the same actions are legal
    np.testing.assert_allclose(res['q'][fin], f['q'][fin], rtol=1e-12, atol=1e-9)  # float64 sums in a different order
    np.testing.assert_allclose(res['value'], f['value'], rtol=1e-12, atol=1e-9)
    # best action: equal, except on exact ties, where the reference-composed float sum decides by rounding noise and the
    # definition here is "the smallest maximising action"
    top = np.where(fin, f['q'], -np.inf).max(axis=1)
    for i in np.flatnonzero(res['best'] != f['best']):
        assert res['best'][i] != 255 and f['best'][i] != 255
        assert abs(f['q'][i, res['best'][i]] - top[i]) <= 1e-9 and res['best'][i] < f['best'][i]
    assert (res['best'] != f['best']).sum() <= 8
    assert np.array_equal(res['best'] == 255, ~fin.any(axis=1))  # terminal boards


def test_oracle_depth2_matches_the_reference_composed_fixture(golden):
    f = golden['expectimax_d2']
    res = oracle.expectimax_depth2(f['boards'])
    check_against_reference_fixture(res, f)
    assert (f['best'] == 255).sum() >= 10 and np.isfinite(f['q']).sum() > 1500  # the fixture covers terminal and ordinary boards


def test_depth2_through_ops(api_runtime):
    """ops.expectimax_depth2 (host logic on the stand-in; the HIP kernel on the GPU) on a handful of hand-checkable boards."""
    # one row [2, 2, 0, 0]: left/right merge to 4 (r = 4); afterwards 15 empty cells
    b = np.zeros((3, 16), dtype=np.uint8)
    b[0, 0] = b[0, 1] = 1
    b[1] = np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 1], dtype=np.uint8)  # checkerboard: terminal
    b[2, 5] = 3  # a lone 8
    t = torch.from_numpy(b).to(api_runtime.device)
    out = ops.expectimax_depth2(t, want_terms=True)
    ref = oracle.expectimax_depth2(b)
    assert np.array_equal(out['terms'].cpu().numpy(), ref['terms'])
    assert np.array_equal(out['q'].cpu().numpy(), ref['q'].astype(np.float32))
    assert np.array_equal(out['best'].cpu().numpy(), ref['best']) and out['best'][1] == 255 and float(out['value'][1]) == 0.0
    assert out['terms'][0, 0, 0] == 4 and out['terms'][0, 0, 3] == 15 and float(out['q'][0, 0]) > 4.0  # a second merge is possible after some spawns
    q2 = out['q'][2].cpu().numpy()
    assert np.all(q2 == 0.0)  # a lone tile: every move is legal, nothing can merge within two plies


@pytest.mark.gpu
def test_depth2_kernel_matches_oracle_and_reference_fixture(golden):
    lib = capi.library()
    stream = lambda: torch.cuda.current_stream(DEV).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    f = golden['expectimax_d2']
    rng = np.random.default_rng(31)
    play = oracle.rollout_batch(oracle.reset_batch(3000, 5, 0)[0], 5, 0, 120, flags=3)['boards']  # mid-game boards of random play
    dense = np.where(rng.random((1500, 16)) < 0.1, 0, rng.integers(1, 6, size=(1500, 16))).astype(np.uint8)
    big = np.where(rng.random((500, 16)) < 0.4, 0, rng.integers(0, 18, size=(500, 16))).astype(np.uint8)
    all_boards = np.concatenate([f['boards'], play, dense, big])
    for n in (1, 63, 64, 65, 257, all_boards.shape[0]):
        boards = all_boards[:n]
        b = torch.from_numpy(boards).cuda()
        terms = torch.full((n, 4, 4), -9, dtype=torch.int32, device='cuda')
        q = torch.full((n, 4), 7.0, dtype=torch.float32, device='cuda')
        best = torch.full((n,), 77, dtype=torch.uint8, device='cuda')
        value = torch.full((n,), 7.0, dtype=torch.float32, device='cuda')
        lib.expectimax_depth2(DEV, stream(), n, p(b), p(terms), p(q), p(best), p(value))
        torch.cuda.synchronize()
        ref = oracle.expectimax_depth2(boards)
        np.testing.assert_array_equal(terms.cpu().numpy(), ref['terms'], err_msg=f'terms n={n}')
        np.testing.assert_array_equal(q.cpu().numpy(), ref['q'].astype(np.float32), err_msg=f'q n={n}')
        np.testing.assert_array_equal(best.cpu().numpy(), ref['best'], err_msg=f'best n={n}')
        np.testing.assert_array_equal(value.cpu().numpy(), ref['value'].astype(np.float32), err_msg=f'value n={n}')
        np.testing.assert_array_equal(b.cpu().numpy(), boards)  # input untouched
        only_best = torch.full((n,), 77, dtype=torch.uint8, device='cuda')  # optional outputs left out
        lib.expectimax_depth2(DEV, stream(), n, p(b), None, None, p(only_best), None)
        torch.cuda.synchronize()
        assert torch.equal(only_best, best)
    got = {'terms': terms.cpu().numpy()[:len(f['boards'])], 'q': q.cpu().numpy()[:len(f['boards'])].astype(np.float64),
           'best': best.cpu().numpy()[:len(f['boards'])], 'value': value.cpu().numpy()[:len(f['boards'])].astype(np.float64)}
    # against the reference-composed fixture directly (q in float32 here)
    np.testing.assert_array_equal(got['terms'], f['terms'])
    fin = np.isfinite(f['q'])
    np.testing.assert_allclose(got['q'][fin], f['q'][fin], rtol=2e-7)
    with pytest.raises(capi.Env2048Error):
        lib.expectimax_depth2(DEV, stream(), 4, p(b), None, None, None, None)  # no output requested
    with pytest.raises(capi.Env2048Error):
        lib.expectimax_depth2(DEV, stream(), 4, p(b) + 1, None, None, p(best), None)  # misaligned boards


def test_opposite_moves_score_the_same(golden):
    """What the depth-2 kernel's leaf evaluation rests on (two slides instead of four): the merge score of a move equals that of
    the opposite move -- on the REFERENCE's own latent_state outputs (fixture afterstates.npz: reward of all four actions of
    6008 boards, written by oracle/gen_golden.py from the imported reference) and on 200 000 random boards through the oracle."""
    r = golden['afterstates']['reward']
    assert np.array_equal(r[:, 0], r[:, 2]) and np.array_equal(r[:, 1], r[:, 3])
    assert (r[:, 0] != r[:, 1]).any()  # (horizontal and vertical scores do differ)
    rng = np.random.default_rng(8)
    n = 200_000
    boards = np.where(rng.random((n, 16)) < 0.25, 0, rng.integers(1, 6, size=(n, 16))).astype(np.uint8)
    boards[n // 2:] = np.where(rng.random((n - n // 2, 16)) < 0.1, 0, rng.integers(0, 18, size=(n - n // 2, 16)))
    score = [oracle.afterstate_batch(boards, np.full(n, a, dtype=np.uint8))['reward'] for a in range(4)]
    assert np.array_equal(score[0], score[2]) and np.array_equal(score[1], score[3])
