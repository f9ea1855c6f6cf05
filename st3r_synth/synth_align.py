"""Synthetic "condensed" global-alignment problems (path B) for parity tests and benchmarks.

Produces exactly the input structure the reference's optimiser consumes
(starster/reconstruct.py:116-127 `sparse_scene_optimizer_slam(imgs, subsample, imsizes, pps,
base_focals, core_depth, anchors, corres, corres2d, preds_21, canonical_paths, mst, ...)`),
as plain numpy arrays in a dict, so that the same problem can be handed to
  - the real reference function (tools/gen_align_goldens.py, this container only),
  - the CPU oracle (oracle/align_oracle.py),
  - the HIP path (starst3r_amd/align.py).
Layout (C views, image W x H, subsample 8 => G = (H/8)*(W/8) core depths per view):
  imsizes int64 [C,2] (W,H) | pps float32 [C,2] pixels | base_focals float32 [C]
  core_depth float32 [C,G] | per view anchors: pixels float32 [n_i,2], idxs int64 [n_i], offsets float32 [n_i]
  pairs: list of (i, j, a_i0, a_j0, n, confs[n]) : n correspondences stored at anchors_i[a_i0:a_i0+n] and
         anchors_j[a_j0:a_j0+n]  (the reference's `slice1`/`slice2`)
  mst: (root, [(i,j), ...]) kinematic chain | preds_21: for non-matching pairs (regression fallback)
"""
import math

import numpy as np


def _look_at_c2w(eye, target=(0, 0, 0), up=(0, 0, 1)):
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64); up = np.asarray(up, np.float64)
    f = target - eye; f /= np.linalg.norm(f)
    r = np.cross(f, up); r /= np.linalg.norm(r)
    d = np.cross(f, r)
    M = np.eye(4)
    M[:3, 0] = r; M[:3, 1] = d; M[:3, 2] = f; M[:3, 3] = eye
    return M


def make_problem(n_views=2, width=512, height=384, n_corr=2000, seed=0, bad_pair=False, noise=0.01, sizes=None):
    """n_corr = correspondences per ordered view pair; bad_pair makes pair (0, C-1) fail the
    matching-confidence gate so the DUSt3R-regression fallback (reconstruct.py:311-323) is exercised.
    sizes: per-view (width, height) -- e.g. one landscape and one portrait photo: the reference keeps per-view lists
    (reconstruct.py:170-177, 276) and accepts them; core_depth is then a LIST of arrays of different lengths."""
    if sizes is not None:
        return _make_problem_mixed(n_views, sizes, n_corr, seed, bad_pair, noise)
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    C, W, H, S = n_views, width, height, 8
    gw, gh = W // S, H // S
    f_true = 1.1 * W
    K_true = np.array([[f_true, 0, W / 2], [0, f_true, H / 2], [0, 0, 1]])
    c2w = [_look_at_c2w((2.5 * math.cos(0.5 * k), 2.5 * math.sin(0.5 * k), 0.3 * math.sin(1.3 * k)))
           for k in range(C)]
    w2c = [np.linalg.inv(m) for m in c2w]

    def depth_of(view, px):
        """depth (camera z) of the unit sphere surface seen at pixels px [n,2] of `view` (nan if missed)."""
        rays = np.stack([(px[:, 0] - W / 2) / f_true, (px[:, 1] - H / 2) / f_true, np.ones(len(px))], -1)
        R, o = c2w[view][:3, :3], c2w[view][:3, 3]
        d = rays @ R.T
        b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
        disc = b * b - a * cc
        t = (-b - np.sqrt(np.maximum(disc, 0))) / a
        t[disc < 0] = np.nan
        return t  # rays have z=1 in camera frame, so t is the depth

    core_depth = np.zeros((C, gh * gw), np.float32)
    for v in range(C):
        gy, gx = np.mgrid[0:gh, 0:gw]
        px = np.stack([gx.reshape(-1) * S + S / 2, gy.reshape(-1) * S + S / 2], -1).astype(np.float64)
        d = depth_of(v, px)
        d[np.isnan(d)] = 3.5  # background plane
        core_depth[v] = (d * (1 + noise * rng.standard_normal(d.shape))).astype(np.float32)

    anchors = [dict(pixels=[], idxs=[], offsets=[]) for _ in range(C)]
    pairs = []
    for i in range(C):
        for j in range(i + 1, C):
            # sample sphere points visible from both cameras
            pts = rng.standard_normal((n_corr * 6, 3)); pts /= np.linalg.norm(pts, axis=1, keepdims=True)
            keep = []
            proj = {}
            for v in (i, j):
                pc = pts @ w2c[v][:3, :3].T + w2c[v][:3, 3]
                uv = np.stack([f_true * pc[:, 0] / pc[:, 2] + W / 2, f_true * pc[:, 1] / pc[:, 2] + H / 2], -1)
                facing = (pts * (c2w[v][:3, 3] - pts)).sum(-1) > 0.05
                inside = (uv[:, 0] > 1) & (uv[:, 0] < W - 2) & (uv[:, 1] > 1) & (uv[:, 1] < H - 2)
                keep.append(facing & inside & (pc[:, 2] > 0.1)); proj[v] = (uv, pc[:, 2])
            ok = np.nonzero(keep[0] & keep[1])[0][:n_corr]
            n = len(ok)
            if n == 0:
                continue  # the two views see disjoint parts of the sphere: no pair, as in a real scene graph
            starts = {}
            for v in (i, j):
                uv, z = proj[v][0][ok], proj[v][1][ok]
                uv = uv + 0.3 * rng.standard_normal(uv.shape)  # matching noise (pixels)
                idx = (np.floor(uv[:, 1] / S).astype(np.int64) * gw + np.floor(uv[:, 0] / S).astype(np.int64))
                off = z / core_depth[v][idx]
                starts[v] = sum(len(a) for a in anchors[v]["idxs"])
                anchors[v]["pixels"].append(uv.astype(np.float32)); anchors[v]["idxs"].append(idx)
                anchors[v]["offsets"].append(off.astype(np.float32))
            confs = rng.uniform(1.0, 12.0, n).astype(np.float32)
            if bad_pair and (i, j) == (0, C - 1):
                confs = rng.uniform(1.0, 4.5, n).astype(np.float32)  # max <= 5: matching gate fails
            pairs.append((i, j, int(starts[i]), int(starts[j]), n, confs))
    for v in range(C):
        a = anchors[v]
        a["pixels"] = np.concatenate(a["pixels"]).astype(np.float32)
        a["idxs"] = np.concatenate(a["idxs"]).astype(np.int64)
        a["offsets"] = np.concatenate(a["offsets"]).astype(np.float32)

    imsizes = np.tile(np.array([[W, H]], np.int64), (C, 1))
    pps = (np.array([[W / 2, H / 2]]) + 3.0 * rng.standard_normal((C, 2))).astype(np.float32)
    base_focals = (f_true * (1 + 0.05 * rng.standard_normal(C))).astype(np.float32)
    mst = (0, [(k, k + 1) for k in range(C - 1)])

    # regression fallback targets for non-matching pairs: points of img1's anchors expressed in img2's camera
    preds_21 = {}
    for (i, j, ai, aj, n, confs) in pairs:
        if confs.max() > 5.0:
            continue
        for (i1, i2) in ((i, j), (j, i)):
            a = anchors[i1]
            z = core_depth[i1][a["idxs"]] * a["offsets"]
            pc1 = np.stack([(a["pixels"][:, 0] - W / 2) / f_true * z, (a["pixels"][:, 1] - H / 2) / f_true * z, z], -1)
            pw = pc1 @ c2w[i1][:3, :3].T + c2w[i1][:3, 3]
            pc2 = pw @ w2c[i2][:3, :3].T + w2c[i2][:3, 3]
            tgt_conf = rng.uniform(1.0, 3.0, len(z)).astype(np.float32)
            preds_21[(i2, i1)] = (pc2.astype(np.float32), tgt_conf)
    return dict(n_views=C, width=W, height=H, subsample=S, imsizes=imsizes, pps=pps, base_focals=base_focals,
                core_depth=core_depth, anchors=anchors, pairs=pairs, mst=mst, preds_21=preds_21,
                c2w_true=np.stack(c2w).astype(np.float32), K_true=K_true.astype(np.float32))


def _make_problem_mixed(n_views, sizes, n_corr, seed, bad_pair, noise):
    """make_problem with a (width, height) per view: same scene (unit sphere), same focal length for every camera."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    C, S = n_views, 8
    assert len(sizes) == C
    Ws = [int(w) for w, _ in sizes]; Hs = [int(h) for _, h in sizes]
    f_true = 1.1 * max(max(Ws), max(Hs))
    c2w = [_look_at_c2w((2.5 * math.cos(0.5 * k), 2.5 * math.sin(0.5 * k), 0.3 * math.sin(1.3 * k))) for k in range(C)]
    w2c = [np.linalg.inv(m) for m in c2w]

    def depth_of(view, px):
        W, H = Ws[view], Hs[view]
        rays = np.stack([(px[:, 0] - W / 2) / f_true, (px[:, 1] - H / 2) / f_true, np.ones(len(px))], -1)
        R, o = c2w[view][:3, :3], c2w[view][:3, 3]
        d = rays @ R.T
        b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
        disc = b * b - a * cc
        t = (-b - np.sqrt(np.maximum(disc, 0))) / a
        t[disc < 0] = np.nan
        return t

    core_depth = []
    for v in range(C):
        gw, gh = Ws[v] // S, Hs[v] // S
        gy, gx = np.mgrid[0:gh, 0:gw]
        px = np.stack([gx.reshape(-1) * S + S / 2, gy.reshape(-1) * S + S / 2], -1).astype(np.float64)
        d = depth_of(v, px)
        d[np.isnan(d)] = 3.5
        core_depth.append((d * (1 + noise * rng.standard_normal(d.shape))).astype(np.float32))
    anchors = [dict(pixels=[], idxs=[], offsets=[]) for _ in range(C)]
    pairs = []
    for i in range(C):
        for j in range(i + 1, C):
            pts = rng.standard_normal((n_corr * 6, 3)); pts /= np.linalg.norm(pts, axis=1, keepdims=True)
            keep, proj = [], {}
            for v in (i, j):
                W, H = Ws[v], Hs[v]
                pc = pts @ w2c[v][:3, :3].T + w2c[v][:3, 3]
                uv = np.stack([f_true * pc[:, 0] / pc[:, 2] + W / 2, f_true * pc[:, 1] / pc[:, 2] + H / 2], -1)
                facing = (pts * (c2w[v][:3, 3] - pts)).sum(-1) > 0.05
                inside = (uv[:, 0] > 1) & (uv[:, 0] < W - 2) & (uv[:, 1] > 1) & (uv[:, 1] < H - 2)
                keep.append(facing & inside & (pc[:, 2] > 0.1)); proj[v] = (uv, pc[:, 2])
            ok = np.nonzero(keep[0] & keep[1])[0][:n_corr]
            n = len(ok)
            if n == 0:
                continue
            starts = {}
            for v in (i, j):
                gw = Ws[v] // S
                uv, z = proj[v][0][ok], proj[v][1][ok]
                uv = uv + 0.3 * rng.standard_normal(uv.shape)
                idx = (np.floor(uv[:, 1] / S).astype(np.int64) * gw + np.floor(uv[:, 0] / S).astype(np.int64))
                off = z / core_depth[v][idx]
                starts[v] = sum(len(a) for a in anchors[v]["idxs"])
                anchors[v]["pixels"].append(uv.astype(np.float32)); anchors[v]["idxs"].append(idx)
                anchors[v]["offsets"].append(off.astype(np.float32))
            confs = rng.uniform(1.0, 12.0, n).astype(np.float32)
            if bad_pair and (i, j) == (0, C - 1):
                confs = rng.uniform(1.0, 4.5, n).astype(np.float32)
            pairs.append((i, j, int(starts[i]), int(starts[j]), n, confs))
    for v in range(C):
        a = anchors[v]
        a["pixels"] = np.concatenate(a["pixels"]).astype(np.float32)
        a["idxs"] = np.concatenate(a["idxs"]).astype(np.int64)
        a["offsets"] = np.concatenate(a["offsets"]).astype(np.float32)
    imsizes = np.array([[Ws[v], Hs[v]] for v in range(C)], np.int64)
    pps = (np.array([[Ws[v] / 2, Hs[v] / 2] for v in range(C)]) + 3.0 * rng.standard_normal((C, 2))).astype(np.float32)
    base_focals = (f_true * (1 + 0.05 * rng.standard_normal(C))).astype(np.float32)
    mst = (0, [(k, k + 1) for k in range(C - 1)])
    preds_21 = {}
    for (i, j, ai, aj, n, confs) in pairs:
        if confs.max() > 5.0:
            continue
        for (i1, i2) in ((i, j), (j, i)):
            a = anchors[i1]
            z = core_depth[i1][a["idxs"]] * a["offsets"]
            pc1 = np.stack([(a["pixels"][:, 0] - Ws[i1] / 2) / f_true * z, (a["pixels"][:, 1] - Hs[i1] / 2) / f_true * z, z], -1)
            pw = pc1 @ c2w[i1][:3, :3].T + c2w[i1][:3, 3]
            pc2 = pw @ w2c[i2][:3, :3].T + w2c[i2][:3, 3]
            preds_21[(i2, i1)] = (pc2.astype(np.float32), rng.uniform(1.0, 3.0, len(z)).astype(np.float32))
    return dict(n_views=C, width=None, height=None, subsample=S, imsizes=imsizes, pps=pps, base_focals=base_focals,
                core_depth=core_depth, anchors=anchors, pairs=pairs, mst=mst, preds_21=preds_21,
                c2w_true=np.stack(c2w).astype(np.float32), f_true=np.float32(f_true))


def pad_core_depth(core_depth):
    """list of per-view core-depth vectors (or a [C, G] array) -> ([C, Gmax] float32 padded with 1, lengths int64 [C])."""
    rows = [np.asarray(d, np.float32).reshape(-1) for d in core_depth]
    lens = np.array([len(r) for r in rows], np.int64)
    out = np.ones((len(rows), int(lens.max())), np.float32)
    for v, r in enumerate(rows):
        out[v, :len(r)] = r
    return out, lens


def flatten(problem):
    """Problem -> flat dict of numpy arrays (npz friendly, and the layout the C ABI consumes):
      anchor arrays concatenated over views with anchor_off [C+1];
      corr arrays (one row per correspondence of every ORDERED matching pair (i,j) and (j,i)):
        corr_a1 / corr_a2 = global anchor indices, corr_conf; (loss_3d, reconstruct.py:325-353)
      corr2d_* per image: pixel in img1, global anchor index of the 3-D point in img2, conf, img1 id
        (loss_2d, reconstruct.py:355-369);
      dust_* : regression fallback rows (global anchor index in img1, target point in cam2, img2, conf)."""
    P = problem
    C = P["n_views"]
    anchor_off = np.zeros(C + 1, np.int64)
    for v in range(C):
        anchor_off[v + 1] = anchor_off[v] + len(P["anchors"][v]["idxs"])
    core, core_len = pad_core_depth(P["core_depth"])   # views of different sizes: rows padded to the longest
    out = dict(n_views=np.int64(C), imsizes=P["imsizes"], pps=P["pps"], base_focals=P["base_focals"],
               core_depth=core, core_len=core_len, anchor_off=anchor_off,
               anchor_pix=np.concatenate([P["anchors"][v]["pixels"] for v in range(C)]),
               anchor_idx=np.concatenate([P["anchors"][v]["idxs"] for v in range(C)]),
               anchor_offset=np.concatenate([P["anchors"][v]["offsets"] for v in range(C)]),
               anchor_img=np.concatenate([np.full(len(P["anchors"][v]["idxs"]), v, np.int32) for v in range(C)]),
               mst_root=np.int64(P["mst"][0]), mst_edges=np.array(P["mst"][1], np.int64).reshape(-1, 2))
    a1, a2, cf = [], [], []
    c2 = {v: dict(pix=[], a2=[], conf=[]) for v in range(C)}
    d_a1, d_tgt, d_img2, d_conf = [], [], [], []
    for (i, j, ai, aj, n, confs) in P["pairs"]:
        gi = anchor_off[i] + ai + np.arange(n); gj = anchor_off[j] + aj + np.arange(n)
        if confs.max() > 5.0:
            a1 += [gi, gj]; a2 += [gj, gi]; cf += [confs, confs]
            c2[i]["pix"].append(P["anchors"][i]["pixels"][ai:ai + n]); c2[i]["a2"].append(gj); c2[i]["conf"].append(confs)
            c2[j]["pix"].append(P["anchors"][j]["pixels"][aj:aj + n]); c2[j]["a2"].append(gi); c2[j]["conf"].append(confs)
        else:
            for (i1, i2) in ((i, j), (j, i)):
                tgt, tc = P["preds_21"][(i2, i1)]
                d_a1.append(anchor_off[i1] + np.arange(len(tc))); d_tgt.append(tgt)
                d_img2.append(np.full(len(tc), i2, np.int32)); d_conf.append(tc)
    cat = lambda xs, dt, shape=(0,): np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dt)
    out.update(corr_a1=cat(a1, np.int64), corr_a2=cat(a2, np.int64), corr_conf=cat(cf, np.float32))
    out.update(c2d_pix=cat([p for v in range(C) for p in c2[v]["pix"]], np.float32, (0, 2)),
               c2d_a2=cat([p for v in range(C) for p in c2[v]["a2"]], np.int64),
               c2d_conf=cat([p for v in range(C) for p in c2[v]["conf"]], np.float32),
               c2d_img1=cat([np.full(sum(len(x) for x in c2[v]["a2"]), v, np.int32) for v in range(C)], np.int32))
    out.update(dust_a1=cat(d_a1, np.int64), dust_tgt=cat(d_tgt, np.float32, (0, 3)), dust_img2=cat(d_img2, np.int32),
               dust_conf=cat(d_conf, np.float32))
    return out


from starst3r_amd.condense import Slice  # noqa: E402  (the product's type; re-exported for the generators' users)


def to_reference_inputs(P):
    """Problem -> the argument objects of the reference's sparse_scene_optimizer_slam(imgs, subsample, imsizes, pps,
    base_focals, core_depth, anchors, corres, corres2d, preds_21, canonical_paths, mst, ...) as torch tensors
    (the structure Mast3r's condense_data produces).  tools/gen_align_goldens.py feeds exactly
    these objects to the reference's own function."""
    import torch
    C = P["n_views"]
    imgs = [f"{i}.png" for i in range(C)]  # the reference feeds fake names (starster/scene.py:120)
    t = torch.tensor
    anchors = {v: (t(P["anchors"][v]["pixels"]), t(P["anchors"][v]["idxs"]), t(P["anchors"][v]["offsets"]))
               for v in range(C)}
    slices = []
    c2d = {v: dict(pix=[], conf=[], sl=[]) for v in range(C)}
    for (i, j, ai, aj, n, confs) in P["pairs"]:
        cf = t(confs)
        slices.append(Slice(i, slice(ai, ai + n), j, slice(aj, aj + n), cf))
        slices.append(Slice(j, slice(aj, aj + n), i, slice(ai, ai + n), cf))
        c2d[i]["pix"].append(anchors[i][0][ai:ai + n]); c2d[i]["conf"].append(cf); c2d[i]["sl"].append((j, slice(aj, aj + n)))
        c2d[j]["pix"].append(anchors[j][0][aj:aj + n]); c2d[j]["conf"].append(cf); c2d[j]["sl"].append((i, slice(ai, ai + n)))
    corres2d = []
    for v in range(C):
        pix = torch.cat(c2d[v]["pix"]); cf = torch.cat(c2d[v]["conf"])
        corres2d.append((v, pix, cf, cf.sum(), c2d[v]["sl"]))
    corres = (None, None, slices)
    preds_21 = {}
    for (i2, i1), (pts, cf) in P["preds_21"].items():
        preds_21.setdefault(imgs[i2], {})[imgs[i1]] = (t(pts), t(cf))
    core_depth = [t(P["core_depth"][v].copy()) for v in range(C)]
    return dict(imgs=imgs, subsample=8, imsizes=t(P["imsizes"].copy()), pps=t(P["pps"].copy()),
                base_focals=t(P["base_focals"].copy()), core_depth=core_depth, anchors=anchors, corres=corres,
                corres2d=corres2d, preds_21=preds_21, canonical_paths=None, mst=P["mst"], cache_path=None)
