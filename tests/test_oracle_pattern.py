"""Oracle delay-pattern bookkeeping vs vectors produced by the reference's Pattern class
(tests/golden/make_golden.py::gold_patterns; codebook_patterns.py:137-285, 390-406)."""
import numpy as np
import pytest

from oracle import pattern_oracle as po

CASES = [(4, 0), (55, 0), (220, 0), (221, 166), (20, 8)]


@pytest.mark.parametrize("T,Tp", CASES)
def test_build_and_revert_match_reference(golden, T, Tp):
    g = golden("patterns.npz")
    k = f"T{T}_p{Tp}"
    codes = g[k + "_codes"].astype(np.int64)
    seq, idx, mask = po.build_sequence(codes, 1024)
    assert np.array_equal(seq, g[k + "_seq"])
    assert np.array_equal(idx, g[k + "_idx"])
    assert np.array_equal(mask, g[k + "_mask"])
    rev, ridx, rmask = po.revert_sequence(g[k + "_filled"].astype(np.int64), T, -1)
    assert np.array_equal(rev, g[k + "_rev"])
    assert np.array_equal(ridx, g[k + "_ridx"])
    assert np.array_equal(rmask, g[k + "_rmask"])
    assert po.first_step_with_timestep(9, T, Tp) == int(g[k + "_first"]) == Tp + 1


@pytest.mark.parametrize("K,T", [(1, 1), (2, 3), (9, 4), (9, 37), (4, 16)])
def test_closed_form_equals_layout_walk(K, T):
    a, am = po.build_indexes(K, T)
    b, bm = po.build_indexes_loops(K, T)
    assert np.array_equal(a, b) and np.array_equal(am, bm)
    for S in (T + K, T + K - 2, max(1, T // 2)):
        a, am = po.revert_indexes(K, T, S)
        b, bm = po.revert_indexes_loops(K, T, S)
        assert np.array_equal(a, b) and np.array_equal(am, bm)


def test_round_trip_full_size():
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 1024, size=(3, 9, 220))
    seq, _, mask = po.build_sequence(codes, 1024)
    assert seq.shape == (3, 9, 229) and (seq[:, ~mask] == 1024).all()
    back, _, m = po.revert_sequence(seq, 220, -1)
    assert m.all() and np.array_equal(back, codes)
