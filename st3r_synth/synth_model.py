"""A synthetic stand-in for the Mast3r front end (BASELINE.json configs[0]: plumbing without weights).

Implements the `model.condense(...)` protocol of starst3r_amd.reconstruct on a textured unit sphere seen by
n cameras: "pairwise predictions" are exact geometry plus noise, so the alignment has a known answer and the
3DGS stage has real images to fit.  Pure numpy; used by tests and examples, never by the hot path itself.
"""
import numpy as np

from . import synth_align


class SyntheticPairwiseModel:
    def __init__(self, width=256, height=192, n_corr=600, seed=0, conf=3.0):
        self.width, self.height, self.n_corr, self.seed, self.conf = width, height, n_corr, seed, conf

    def condense(self, imgs, filelist, device, cache_dir):
        C = len(imgs)
        W, H = self.width, self.height
        P = synth_align.make_problem(n_views=C, width=W, height=H, n_corr=self.n_corr, seed=self.seed)
        flat = synth_align.flatten(P)
        f = float(P["K_true"][0, 0])
        S = 8
        gw = W // S
        ys, xs = np.mgrid[0:H, 0:W]
        pix = np.stack([xs.reshape(-1) + 0.5, ys.reshape(-1) + 0.5], -1).astype(np.float64)
        out_imgs, dense = [], []
        for v in range(C):
            c2w = P["c2w_true"][v].astype(np.float64)
            rays = np.stack([(pix[:, 0] - W / 2) / f, (pix[:, 1] - H / 2) / f, np.ones(len(pix))], -1)
            d = rays @ c2w[:3, :3].T; o = c2w[:3, 3]
            b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
            disc = b * b - a * cc
            hit = disc > 0
            t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / a, 3.5)
            pw = o + d * t[:, None]
            # procedural texture of the sphere surface (smooth, view independent), grey background
            tex = 0.5 + 0.5 * np.stack([np.sin(4 * pw[:, 0] + 1), np.sin(5 * pw[:, 1] + 2), np.sin(3 * pw[:, 2])], -1)
            img = np.where(hit[:, None], tex, 0.35).reshape(H, W, 3).astype(np.float32)
            out_imgs.append(img)
            idx = (np.floor(pix[:, 1] / S).astype(np.int64) * gw + np.floor(pix[:, 0] / S).astype(np.int64))
            off = (t / P["core_depth"][v][idx]).astype(np.float32)
            confs = np.where(hit, self.conf, 1.0).astype(np.float32)  # only sphere pixels pass conf_thres=1.5
            dense.append(dict(pixels=pix.astype(np.float32), idxs=idx, offsets=off, confs=confs,
                              base_focal=float(P["base_focals"][v])))
        flat["imgs"] = out_imgs
        flat["dense"] = dense
        return flat


class SyntheticPairModel:
    """The other model protocol of starst3r_amd.reconstruct: `forward_pairs` hands over what Mast3r's forward_mast3r
    caches per image pair (pointmaps, confidences, correspondences -- st3r_synth.synth_pairs) and leaves the
    condensation (starst3r_amd.condense) and the alignment to the library."""
    subsample = 8

    def __init__(self, width=256, height=192, n_corr=1500, seed=0):
        self.width, self.height, self.n_corr, self.seed = width, height, n_corr, seed

    def forward_pairs(self, imgs, filelist, device, cache_dir):
        from . import synth_pairs
        W, H = self.width, self.height
        P = synth_pairs.make_pair_predictions(len(imgs), W, H, self.subsample, seed=self.seed, n_corr=self.n_corr)
        names = list(filelist)
        ren = dict(zip(P["imgs"], names))
        tmp_pairs = {(ren[a], ren[b]): v for (a, b), v in P["pairs"].items()}
        f = P["focal_true"]
        ys, xs = np.mgrid[0:H, 0:W]
        rays = np.stack([(xs - W / 2) / f, (ys - H / 2) / f, np.ones((H, W))], -1).reshape(-1, 3)
        images = []
        for v in range(len(imgs)):
            c2w = P["c2w_true"][v].astype(np.float64)
            d = rays @ c2w[:3, :3].T; o = c2w[:3, 3]
            b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
            disc = b * b - a * cc
            hit = disc > 0
            t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / a, 3.5)
            pw = o + d * t[:, None]
            tex = 0.5 + 0.5 * np.stack([np.sin(4 * pw[:, 0] + 1), np.sin(5 * pw[:, 1] + 2), np.sin(3 * pw[:, 2])], -1)
            images.append(np.where(hit[:, None], tex, 0.35).reshape(H, W, 3).astype(np.float32))
        return tmp_pairs, images


class SyntheticNetwork:
    """Stands in for the Mast3r ViT only: `symmetric_inference` returns the four head outputs of a pair (pointmaps,
    confidences, 24-d descriptors, descriptor confidences) for the synthetic sphere scene.  Matching, the pair cache,
    condensation, alignment and everything after run in the library (starst3r_amd.forward / condense / align)."""
    subsample = 8

    def __init__(self, n_views=4, width=128, height=96, seed=0, noise=0.003):
        from . import synth_pairs
        self.P = synth_pairs.make_pair_predictions(n_views, width, height, self.subsample, seed=seed, noise=noise,
                                                   n_corr=1)
        self.W, self.H, self.noise = width, height, noise
        self.calls = 0
        rng = np.random.Generator(np.random.PCG64(77 + seed))
        self.freq = rng.standard_normal((3, 24)) * 6.0
        self.phase = rng.uniform(0, 2 * np.pi, 24)
        f = self.P["focal_true"]
        ys, xs = np.mgrid[0:height, 0:width]
        self.rays = np.stack([(xs - width / 2) / f, (ys - height / 2) / f, np.ones((height, width))], -1)

    def _view(self, v):
        c2w = self.P["c2w_true"][v].astype(np.float64)
        d = self.rays @ c2w[:3, :3].T; o = c2w[:3, 3]
        b = d @ o; a = (d * d).sum(-1); cc = o @ o - 1.0
        disc = b * b - a * cc
        hit = disc > 0
        t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / a, 3.5)
        return self.rays * t[..., None], o + d * t[..., None], hit, c2w

    def images(self):
        """the input views as (3,H,W) tensors in [-1,1] (what load_images returns)."""
        import torch
        out = []
        for v in range(len(self.P["imgs"])):
            _, pw, hit, _ = self._view(v)
            tex = 0.5 + 0.5 * np.stack([np.sin(4 * pw[..., 0] + 1), np.sin(5 * pw[..., 1] + 2), np.sin(3 * pw[..., 2])], -1)
            img = np.where(hit[..., None], tex, 0.35).astype(np.float32)
            out.append(torch.tensor(img).permute(2, 0, 1) * 2 - 1)
        return out

    def symmetric_inference(self, img1, img2, device):
        import torch
        self.calls += 1
        i, j = int(img1["idx"]), int(img2["idx"])
        rng = np.random.Generator(np.random.PCG64(9000 + 97 * i + j))
        out = {}
        data = {v: self._view(v) for v in (i, j)}

        def head(src, frame):
            pc, pw, hit, _ = data[src]
            w2c = np.linalg.inv(data[frame][3])
            X = pw @ w2c[:3, :3].T + w2c[:3, 3]
            X = X * (1 + self.noise * rng.standard_normal(X.shape[:2] + (1,)))
            conf = 1.0 + np.where(hit, 8.0, 0.5) * rng.uniform(0.5, 1.0, hit.shape)
            desc = np.sin(pw @ self.freq + self.phase)
            desc /= np.linalg.norm(desc, axis=-1, keepdims=True)
            dconf = np.where(hit, 8.0, 0.2) * rng.uniform(0.8, 1.0, hit.shape)
            t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)[None]
            return dict(pts3d=t(X), conf=t(conf), desc=t(desc), desc_conf=t(dconf))
        return head(i, i), head(j, i), head(j, j), head(i, j)
