"""Synthetic pairwise predictions (what Mast3r's forward_mast3r leaves in its cache, starster/reconstruct.py:97) for
the unit-sphere scene of synth_align: for every unordered view pair the four pointmaps X11, X21 (frame of view 1),
X22, X12 (frame of view 2), their confidences, and pixel correspondences.  Used to drive the condensation
(starst3r_amd/condense.py) and, through it, the alignment from "raw" pair data in tests and examples."""
import math

import numpy as np

from .synth_align import _look_at_c2w


def make_pair_predictions(n_views=3, width=256, height=192, subsample=8, seed=0, noise=0.003, n_corr=1500):
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    C, W, H = n_views, width, height
    f_true = 1.1 * W
    c2w = [_look_at_c2w((2.5 * math.cos(0.45 * k), 2.5 * math.sin(0.45 * k), 0.3 * math.sin(1.3 * k))) for k in range(C)]
    w2c = [np.linalg.inv(m) for m in c2w]
    ys, xs = np.mgrid[0:H, 0:W]
    rays = np.stack([(xs - W / 2) / f_true, (ys - H / 2) / f_true, np.ones((H, W))], -1)   # pixel grid = xy_grid(W, H)

    def depth_of(v):
        R, o = c2w[v][:3, :3], c2w[v][:3, 3]
        d = rays @ R.T
        b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
        disc = b * b - a * cc
        t = (-b - np.sqrt(np.maximum(disc, 0))) / a
        hit = disc > 0
        t = np.where(hit, t, 3.5)      # background plane at depth 3.5
        return t, hit
    depth = {}; hit = {}; cam_pts = {}
    for v in range(C):
        depth[v], hit[v] = depth_of(v)
        cam_pts[v] = rays * depth[v][..., None]

    def to_frame(pts_cam, src, dst):
        pw = pts_cam @ c2w[src][:3, :3].T + c2w[src][:3, 3]
        return pw @ w2c[dst][:3, :3].T + w2c[dst][:3, 3]

    def noisy(X):
        return (X * (1 + noise * rng.standard_normal(X.shape[:2] + (1,)))).astype(np.float32)

    def conf(v):
        return (1.0 + np.where(hit[v], 8.0, 0.5) * rng.uniform(0.5, 1.0, (H, W))).astype(np.float32)
    imgs = [f"{i}.png" for i in range(C)]
    pairs = {}
    for i in range(C):
        for j in range(i + 1, C):
            pred1 = (noisy(cam_pts[i]), conf(i), noisy(to_frame(cam_pts[j], j, i)), conf(j))   # X11 C11 X21 C21
            pred2 = (noisy(cam_pts[j]), conf(j), noisy(to_frame(cam_pts[i], i, j)), conf(i))   # X22 C22 X12 C12
            # correspondences: sphere pixels of i re-projected into j, visible from j
            cand = np.stack([rng.integers(2, W - 2, n_corr * 4), rng.integers(2, H - 2, n_corr * 4)], -1)
            ok = hit[i][cand[:, 1], cand[:, 0]]
            cand = cand[ok]
            pc_i = cam_pts[i][cand[:, 1], cand[:, 0]]
            pw = pc_i @ c2w[i][:3, :3].T + c2w[i][:3, 3]
            pc_j = pw @ w2c[j][:3, :3].T + w2c[j][:3, 3]
            uv = np.stack([f_true * pc_j[:, 0] / pc_j[:, 2] + W / 2, f_true * pc_j[:, 1] / pc_j[:, 2] + H / 2], -1)
            facing = (pw * (c2w[j][:3, 3] - pw)).sum(-1) > 0.05
            inside = (uv[:, 0] > 1) & (uv[:, 0] < W - 2) & (uv[:, 1] > 1) & (uv[:, 1] < H - 2) & (pc_j[:, 2] > 0.1)
            keep = np.nonzero(facing & inside)[0][:n_corr]
            xy1 = cand[keep].astype(np.float32)
            xy2 = np.round(uv[keep]).astype(np.float32)       # matches live on the pixel grid (fast_reciprocal_NNs)
            cf = rng.uniform(1.0, 12.0, len(keep)).astype(np.float32)
            score = (float(np.sqrt(np.sqrt(pred1[1].mean() * pred1[3].mean() * pred2[1].mean() * pred2[3].mean()))),
                     float(cf.sum()), len(cf))
            pairs[(imgs[i], imgs[j])] = ((pred1, pred2), (score, (xy1, xy2, cf)))
    return dict(imgs=imgs, pairs=pairs, subsample=subsample, width=W, height=H, focal_true=f_true,
                c2w_true=np.stack(c2w).astype(np.float32))
