"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path) for the canonical-data condensation,
SURVEY 8(f) row 2: what starster/reconstruct.py:101-106 calls in Mast3r (prepare_canonical_data,
compute_min_spanning_tree, condense_data -> mast3r/cloud_opt/sparse_ga.py, and dust3r's
estimate_focal_knowing_depth).  The mast3r submodule is NOT vendored under /root/reference (empty directory), so
this file restates the upstream algorithm from its published source and is marked [U] like SURVEY Appendix A:
PARITY UNPINNED against upstream -- the HIP kernels are pinned against THIS restatement, and the whole chain against
geometric ground truth of a synthetic scene (tests/test_gpu_condense.py).  float32 arithmetic like upstream's torch code.
"""
import numpy as np

F = np.float32
EPS = np.finfo(np.float32).eps


def canonical_view(ptmaps11, confs11, subsample, mode="avg-angle"):
    """ptmaps11 [n,H,W,3], confs11 [n,H,W] -> canon [H,W,3], canon2 [H,W], cconf [H,W] (canonical_view [U])."""
    assert mode == "avg-angle"
    X = np.asarray(ptmaps11, F); Cf = np.asarray(confs11, F)
    n, H, W, _ = X.shape
    w = (Cf - F(0.999))[..., None]
    sw = w.sum(0, dtype=F)
    canon = (w * X).sum(0, dtype=F) / sw
    S = subsample
    cy = (np.arange(H) // S) * S + S // 2; cx = (np.arange(W) // S) * S + S // 2
    Xc = X[:, cy][:, :, cx]                                   # block-centre prediction of every pixel's block
    zc = np.maximum(Xc[..., 2], EPS)
    r = np.maximum(np.sqrt((X[..., 0] - Xc[..., 0]) ** 2 + (X[..., 1] - Xc[..., 1]) ** 2, dtype=F), F(1e-8))
    ang = np.arctan((X[..., 2] - zc) / r).astype(F)
    avg = (w[..., 0] * ang).sum(0, dtype=F) / sw[..., 0]
    depth = (r.sum(0, dtype=F) / F(n)) * np.tan(avg).astype(F)
    canon2 = F(1) + depth / canon[cy][:, cx][..., 2]
    cconf = (w * w).sum(0, dtype=F)[..., 0] / sw[..., 0]
    return canon.astype(F), canon2.astype(F), cconf.astype(F)


def estimate_focal_knowing_depth(canon, pp, min_focal=0.5, max_focal=3.5):
    """dust3r post_process.estimate_focal_knowing_depth(focal_mode='weiszfeld') [U]; canon [H,W,3], pp (x, y)."""
    H, W, _ = canon.shape
    ys, xs = np.mgrid[0:H, 0:W]
    px = np.stack([xs - F(pp[0]), ys - F(pp[1])], -1).reshape(-1, 2).astype(F)
    P = canon.reshape(-1, 3).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        xyz = P[:, :2] / P[:, 2:3]
    xyz = np.nan_to_num(xyz, nan=0.0, posinf=0.0, neginf=0.0).astype(F)
    dpx = (xyz * px).sum(-1); dxx = (xyz * xyz).sum(-1)
    focal = F(np.mean(dpx, dtype=np.float64) / np.mean(dxx, dtype=np.float64))
    for _ in range(10):
        dis = np.sqrt(((px - focal * xyz) ** 2).sum(-1))
        w = 1.0 / np.maximum(dis, 1e-8)
        focal = F(np.mean(w * dpx, dtype=np.float64) / np.mean(w * dxx, dtype=np.float64))
    base = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
    return F(np.clip(focal, min_focal * base, max_focal * base))


def anchor_depth_offsets(canon2, xy, subsample):
    """anchor_depth_offsets [U] for one pixel list xy [n,2]: (core index, relative depth offset)."""
    H, W = canon2.shape
    S = subsample
    W2 = len(range(S // 2, W, S))
    px = xy[:, 0].astype(np.int64); py = xy[:, 1].astype(np.int64)
    idx = (py // S) * W2 + (px // S)
    ref = canon2[(py // S) * S + S // 2, (px // S) * S + S // 2]
    return idx, (canon2[py, px] / ref).astype(F)


def compute_min_spanning_tree(pws):
    """compute_min_spanning_tree [U]: maximum-score spanning tree of the symmetric score matrix, rooted at the end of
    its best edge that has the larger total score, edges in breadth-first order as (parent, child)."""
    pws = np.asarray(pws, np.float64)
    C = pws.shape[0]
    iu = [(pws[i, j], i, j) for i in range(C) for j in range(i + 1, C) if pws[i, j] > 0]
    iu.sort(key=lambda e: (-e[0], e[1], e[2]))
    parent = list(range(C))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    adj = {i: [] for i in range(C)}
    for s, i, j in iu:
        ri, rj = find(i), find(j)
        if ri != rj:
            parent[ri] = rj; adj[i].append(j); adj[j].append(i)
    if not iu:
        return 0, []
    _, bi, bj = iu[0]
    root = bi if pws[bi].sum() >= pws[bj].sum() else bj
    seen = {root}; order = [root]; edges = []
    for a in order:
        for b in sorted(adj[a]):
            if b not in seen:
                seen.add(b); order.append(b); edges.append((a, b))
    return root, edges
