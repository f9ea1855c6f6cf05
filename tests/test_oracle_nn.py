"""Pins oracle/nn_oracle.py (path A restatement, "parity unpinned vs upstream") -- CPU only."""
import numpy as np

from oracle import nn_oracle as no


def test_nn_dot_is_bruteforce_argmax():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((50, 24)).astype(np.float32); d = rng.standard_normal((3000, 24)).astype(np.float32)
    idx, best, second = no.nn_dot(q, d, block=512, dtype=np.float64)
    s = q.astype(np.float64) @ d.astype(np.float64).T
    assert np.array_equal(idx, s.argmax(1))
    np.testing.assert_allclose(best, s.max(1))
    np.testing.assert_allclose(second, np.sort(s, 1)[:, -2])


def test_first_index_wins_ties():
    d = np.zeros((10, 24), np.float32); d[3, 0] = 1; d[7, 0] = 1
    q = np.zeros((1, 24), np.float32); q[0, 0] = 2
    assert no.nn_dot(q, d, block=4)[0][0] == 3


def test_reciprocal_matches_are_mutual_and_recover_planted_pairs():
    H, W = 48, 64
    A, B, src, dst = no.synth_descriptors(H, W, planted=0.2, seed=1)
    xy1, xy2 = no.fast_reciprocal_NNs(A, B, S=4)
    Af, Bf = A.reshape(-1, 24).astype(np.float64), B.reshape(-1, 24).astype(np.float64)
    # every returned pair is a mutual nearest neighbour
    assert np.array_equal((Af[xy1] @ Bf.T).argmax(1), xy2)
    assert np.array_equal((Bf[xy2] @ Af.T).argmax(1), xy1)
    # sorted, unique
    key = xy2.astype(np.int64) | (xy1.astype(np.int64) << 32)
    assert np.all(np.diff(key) > 0)
    planted = dict(zip(src.tolist(), dst.tolist()))
    hits = sum(1 for a, b in zip(xy1.tolist(), xy2.tolist()) if planted.get(a) == b)
    assert hits > 0.25 * len(xy1) and hits >= 20  # random descriptors also produce some mutual pairs
