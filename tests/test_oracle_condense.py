"""CPU checks of the condensation oracle (oracle/condense_oracle.py).  Upstream (mast3r) is not in
the reference tree, so these pin the restatement against closed-form properties, not against upstream outputs."""
import itertools

import numpy as np

from oracle import condense_oracle as co
from st3r_synth import synth_pairs


def _maps_of(P, img):
    pt, cf = [], []
    for (a, b), ((p1, p2), _c) in P["pairs"].items():
        if a == img:
            pt.append(p1[0]); cf.append(p1[1])
        elif b == img:
            pt.append(p2[0]); cf.append(p2[1])
    return np.stack(pt), np.stack(cf)


def test_canonical_view_of_identical_maps_is_the_map():
    P = synth_pairs.make_pair_predictions(3, 64, 48, seed=0, noise=0.0, n_corr=50)
    X, Cf = _maps_of(P, "1.png")
    X = np.repeat(X[:1], 3, 0)
    canon, canon2, cconf = co.canonical_view(X, Cf[[0, 1, 1]], 8)
    assert np.allclose(canon, X[0], rtol=2e-6, atol=1e-6)
    # block centres sit at relative depth 1
    assert np.allclose(canon2[4::8, 4::8], 1.0, atol=1e-6)
    # a pixel's relative depth reproduces its own depth (one prediction: the angle round-trips exactly)
    zc = np.repeat(np.repeat(X[0][4::8, 4::8, 2], 8, 0), 8, 1)
    assert np.allclose(canon2, 1 + (X[0][..., 2] - zc) / zc, rtol=1e-4, atol=1e-5)
    w = Cf[[0, 1, 1]] - np.float32(0.999)
    assert np.allclose(cconf, (w * w).sum(0) / w.sum(0), rtol=1e-6)


def test_focal_is_recovered_and_clipped():
    P = synth_pairs.make_pair_predictions(2, 128, 96, seed=2, noise=0.0, n_corr=50)
    X, Cf = _maps_of(P, "0.png")
    canon, _, _ = co.canonical_view(X, Cf, 8)
    f = co.estimate_focal_knowing_depth(canon, (64, 48))
    assert abs(f - P["focal_true"]) < 1e-3 * P["focal_true"]
    base = 128 / (2 * np.tan(np.deg2rad(30)))
    assert co.estimate_focal_knowing_depth(canon * np.float32([10, 10, 1]), (64, 48)) == np.float32(0.5 * base)
    assert co.estimate_focal_knowing_depth(canon * np.float32([0.01, 0.01, 1]), (64, 48)) == np.float32(3.5 * base)


def test_anchor_offsets():
    rng = np.random.default_rng(0)
    canon2 = rng.uniform(0.5, 2.0, (48, 64)).astype(np.float32)
    xy = np.stack([rng.integers(0, 64, 200), rng.integers(0, 48, 200)], -1).astype(np.float32)
    idx, off = co.anchor_depth_offsets(canon2, xy, 8)
    for k in range(200):
        x, y = int(xy[k, 0]), int(xy[k, 1])
        assert idx[k] == (y // 8) * 8 + x // 8
        assert off[k] == canon2[y, x] / canon2[(y // 8) * 8 + 4, (x // 8) * 8 + 4]


def test_spanning_tree_is_maximal():
    rng = np.random.default_rng(1)
    for C in (2, 4, 6):
        s = rng.uniform(1, 100, (C, C)); s = np.triu(s, 1); s = s + s.T
        root, edges = co.compute_min_spanning_tree(s)
        assert len(edges) == C - 1
        seen = {root}
        for a, b in edges:          # breadth-first (parent, child): every parent is already placed
            assert a in seen and b not in seen
            seen.add(b)
        total = sum(s[a, b] for a, b in edges)
        best = 0
        all_e = [(i, j) for i in range(C) for j in range(i + 1, C)]
        for sub in itertools.combinations(all_e, C - 1):
            comp = list(range(C))

            def find(a):
                while comp[a] != a:
                    a = comp[a]
                return a
            okay = True
            for i, j in sub:
                ri, rj = find(i), find(j)
                if ri == rj:
                    okay = False; break
                comp[ri] = rj
            if okay:
                best = max(best, sum(s[i, j] for i, j in sub))
        assert abs(total - best) < 1e-9


def test_spanning_tree_against_scipy():
    """upstream calls scipy.sparse.csgraph.minimum_spanning_tree on the negated scores: same total weight, a tree."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import minimum_spanning_tree
    rng = np.random.default_rng(3)
    for C in (3, 8, 32):
        s = rng.uniform(1, 100, (C, C)); s = np.triu(s, 1)
        tree = minimum_spanning_tree(sp.csr_matrix(-s))
        root, edges = co.compute_min_spanning_tree(s + s.T)
        assert len(edges) == C - 1 == tree.nnz
        assert abs(sum((s + s.T)[a, b] for a, b in edges) + tree.sum()) < 1e-9
