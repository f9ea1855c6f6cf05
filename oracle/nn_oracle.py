"""CPU restatement of Mast3r's reciprocal-NN matching (path A) -- TEST INFRASTRUCTURE ONLY.

PARITY STATUS: "parity unpinned vs upstream".  The arithmetic lives in the git submodule naver/mast3r
(/root/reference/.gitmodules:1-3), whose directory is empty here and whose pinned commit is unknown; this
file restates `mast3r.fast_nn.fast_reciprocal_NNs` / `bruteforce_reciprocal_nns(dist='dot')` / `merge_corres`
from SURVEY.md App. A.4 as reached from starster/reconstruct.py:97.  It is pinned by brute-force float64
all-pairs checks (tests/test_oracle_nn.py).
"""
import numpy as np


def nn_dot(queries, db, block=8192, dtype=np.float32):
    """argmax_j q.db_j with first-index tie breaking; also returns best and runner-up scores (float64)
    so that tests can recognise near ties."""
    q = queries.astype(dtype); d = db.astype(dtype)
    n = q.shape[0]
    best = np.full(n, -np.inf, np.float64); second = np.full(n, -np.inf, np.float64); idx = np.zeros(n, np.int64)
    for b0 in range(0, d.shape[0], block):
        s = (q @ d[b0:b0 + block].T).astype(np.float64)
        j = s.argmax(1); v = s[np.arange(n), j]
        s2 = s.copy(); s2[np.arange(n), j] = -np.inf
        v2 = s2.max(1) if s.shape[1] > 1 else np.full(n, -np.inf)
        upd = v > best
        second = np.where(upd, np.maximum(best, v2), np.maximum(second, v))
        idx = np.where(upd, b0 + j, idx); best = np.where(upd, v, best)
    return idx.astype(np.int32), best, second


def merge_corres(idx1, idx2):
    c = np.unique(np.c_[idx2, idx1].astype(np.int32).view(np.int64))
    xy2, xy1 = c[:, None].view(np.int32).T
    return xy1, xy2


def fast_reciprocal_NNs(pts1, pts2, S=8, max_iter=10, dtype=np.float64):
    H1, W1, D = pts1.shape; H2, W2, _ = pts2.shape
    A = pts1.reshape(-1, D); B = pts2.reshape(-1, D)
    y1, x1 = np.mgrid[S // 2:H1:S, S // 2:W1:S].reshape(2, -1)
    xy1 = np.int32(np.unique(x1 + W1 * y1)); xy2 = np.full_like(xy1, -1)
    old1, old2 = xy1.copy(), xy2.copy()
    notyet = np.ones(len(xy1), bool)
    it = 0
    while notyet.any():
        xy2[notyet] = nn_dot(A[xy1[notyet]], B, dtype=dtype)[0]
        notyet &= (old2 != xy2)
        xy1[notyet] = nn_dot(B[xy2[notyet]], A, dtype=dtype)[0]
        notyet &= (old1 != xy1)
        it += 1
        if it >= max_iter:
            break
        old2[:] = xy2; old1[:] = xy1
    conv = ~notyet
    return merge_corres(xy1[conv], xy2[conv])


def synth_descriptors(H, W, D=24, planted=0.1, seed=0, noise=0.05):
    """two descriptor maps: random unit vectors; a fraction of image-1 pixels is planted into image 2
    at a shifted location (plus noise) so that true reciprocal matches exist."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((H, W, D)); A /= np.linalg.norm(A, axis=-1, keepdims=True)
    B = rng.standard_normal((H, W, D)); B /= np.linalg.norm(B, axis=-1, keepdims=True)
    n = int(planted * H * W)
    src = rng.choice(H * W, n, replace=False); dst = rng.choice(H * W, n, replace=False)
    Bf = B.reshape(-1, D); Af = A.reshape(-1, D)
    v = Af[src] + noise * rng.standard_normal((n, D)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
    Bf[dst] = v
    return A.astype(np.float32), Bf.reshape(H, W, D).astype(np.float32), src, dst
