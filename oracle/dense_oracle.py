"""CPU restatement (numpy) of the dense point extraction between alignment and 3DGS seeding -- TEST
INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

What it restates (reference call site starster/scene.py:148, `scene.get_dense_pts3d(clean_depth=True)`):
  * Mast3r SparseGA.get_dense_pts3d -> make_dense_pts3d: every pixel is an anchor of its view's optimised core
    depthmap; make_pts3d as in SURVEY.md App. A.5: offset' = 1 + (offset - 1) base_focal / focal,
    z = depthmap[idx] * offset', p_cam = z K^-1 (u, v, 1), p_world = cam2w p_cam;
  * dust3r clean_pointcloud(confs, K, world2cam, depthmaps, pts3d, tol=0.001, bad_conf=0): for i, for j != i
    (confidences updated in place in that order): project view i's points into view j, round to the nearest
    pixel, and where the point is inside the image, in front of j's depth (proj_depth < (1 - tol) depth_j) and
    less confident than that pixel, clip its confidence to bad_conf.

PARITY UNPINNED: mast3r / dust3r are an empty submodule in the reference tree (commit unknown) and the
reference holds no test or fixture for this path; the two functions are restated from their published
behaviour [U].
"""
import numpy as np


def unproject(pixels, idxs, offsets, depthmap, K, cam2w, base_focal):
    """One view. pixels [n,2], idxs [n], offsets [n], depthmap [G] -> (world points [n,3], z [n]) in float64."""
    f = float(K[0, 0])
    offp = 1.0 + (np.asarray(offsets, np.float64) - 1.0) * (float(base_focal) / f)
    z = np.asarray(depthmap, np.float64)[idxs] * offp
    pc = np.stack([(pixels[:, 0] - K[0, 2]) / f * z, (pixels[:, 1] - K[1, 2]) / f * z, z], -1)
    return pc @ np.asarray(cam2w, np.float64)[:3, :3].T + np.asarray(cam2w, np.float64)[:3, 3], z


def clean_pointcloud(confs, Ks, cam2ws, zmaps, pts, sizes, tol=0.001, bad_conf=0.0, margin=None):
    """confs / zmaps / pts: per-view lists ([H*W], [H*W], [H*W,3]); sizes: per-view (H, W).
    Returns the cleaned confidences.  `margin` (optional list) receives, per point, how close any of the float
    decisions came to flipping (pixel rounding, depth test) so that parity tests can skip knife-edge points."""
    res = [np.array(c, np.float64) for c in confs]
    C = len(res)
    w2c = [np.linalg.inv(np.asarray(m, np.float64)) for m in cam2ws]
    for i in range(C):
        m_i = np.full(len(res[i]), np.inf)
        for j in range(C):
            if i == j:
                continue
            Hj, Wj = sizes[j]
            p = pts[i] @ w2c[j][:3, :3].T + w2c[j][:3, 3]
            z = p[:, 2]
            K = np.asarray(Ks[j], np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                uf = (K[0, 0] * p[:, 0] + K[0, 2] * z) / z
                vf = (K[1, 1] * p[:, 1] + K[1, 2] * z) / z
            u = np.rint(uf); v = np.rint(vf)
            inside = (z > 0) & (u >= 0) & (u < Wj) & (v >= 0) & (v < Hj)
            k = (v[inside].astype(np.int64) * Wj + u[inside].astype(np.int64))
            zj = np.asarray(zmaps[j], np.float64)[k]
            bad = (z[inside] < (1 - tol) * zj) & (res[i][inside] < res[j][k])
            idx = np.nonzero(inside)[0]
            res[i][idx[bad]] = np.minimum(res[i][idx[bad]], bad_conf)
            # decision margins: distance of u, v from a rounding boundary or an image border (pixels), of the depth test
            # (relative), of z from 0
            fr = lambda a: np.abs(np.abs(a - np.floor(a)) - 0.5)
            mm = np.minimum(fr(uf), fr(vf))
            mm = np.minimum(mm, np.abs(z) * 1e3)
            dm = np.full(len(z), np.inf)
            dm[idx] = np.abs(z[inside] / ((1 - tol) * zj) - 1.0) * 1e3
            m_i = np.minimum(m_i, np.minimum(np.where(np.isfinite(mm), mm, 0.0), dm))
        if margin is not None:
            margin.append(m_i)
    return res
