"""Reconstruction pipeline -- drop-in for the reference's `starster.reconstruct`
(starster/reconstruct.py:19-113): `reconstruct_scene(model, imgs, filelist, device, optim_params, tmpdir)`
returns the TUPLE (scene, optim_params) exactly like the reference (its docstring says "scene", the code and
its only caller use a tuple: reconstruct.py:72,113; scene.py:122).

What runs where:
  * pairwise inference (the Mast3r ViT) stays on PyTorch-ROCm and is supplied by `model`;
  * reciprocal-NN matching (path A) -> starst3r_amd.matching (MFMA kernel);
  * global alignment (path B)       -> starst3r_amd.align   (fused HIP optimiser);
  * the condensation between them (canonical pointmaps, MST, anchors; Mast3r prepare_canonical_data /
    condense_data) is a "next" row: `model` must provide it for now, see below.

Model protocol.  The Mast3r package is not vendored by the reference (empty submodule) and its weights
cannot be fetched offline, so `model` is any object with
      model.condense(imgs, filelist, device, cache_dir) -> dict
returning the condensed problem in starst3r_amd.synth_align.flatten() layout plus
      "imgs": list of HxWx3 float arrays in [0,1] (the Mast3r-resized GT images),
      "dense": optional per-view dict(pixels [n,2], idxs [n], offsets [n], confs [n], base_focal) -- the dense
               pixels as anchors of the view's core depthmap (dense unprojection).
starst3r_amd.synth_model.SyntheticPairwiseModel implements it on synthetic scenes (BASELINE configs[0]).
"""
__all__ = ("reconstruct_scene", "reconstruct")

import tempfile

import numpy as np
import torch

from . import align


class SparseGAResult:
    """The members of Mast3r's SparseGA that the reference touches (scene.py:133,138-139,148):
    .imgs, .cam2w, .intrinsics, .get_dense_pts3d(clean_depth=True) -> (pts list, depthmaps list, confs list)."""

    def __init__(self, imgs, res, dense=None):
        self.imgs = imgs
        self.cam2w = res["cam2w"]
        self.intrinsics = res["intrinsics"]
        self.depthmaps = res["depthmaps"]
        self.pts3d = res["pts3d"]
        self.losses = res["losses"]
        self._res = res
        self._dense = dense

    def get_dense_pts3d(self, clean_depth=True):
        """Dense unprojection with the OPTIMISED cameras and depthmaps (Mast3r SparseGA.get_dense_pts3d [U]): every
        dense pixel is treated like an anchor -- depth = depthmap[idx] * offset' -- so the points live in the
        optimiser's gauge (App. A.5 make_pts3d); with clean_depth the confidences of points floating in front of
        another view's surface are lowered (dust3r clean_pointcloud [U]).  Runs in libst3r_hip.so
        (st3r_dense_unproject / st3r_dense_clean).  Returns (pts list, depthmaps list, confs list) per view."""
        if self._dense is None:
            raise NotImplementedError("dense unprojection needs the model's dense pixel table (SURVEY.md 8(f) #3)")
        from . import ops
        dev = self.cam2w.device
        ctx = ops.get_context(dev)
        counts = [len(d["idxs"]) for d in self._dense]
        start = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=dev)
        cat = lambda key, dt: torch.cat([torch.as_tensor(np.asarray(d[key]), dtype=dt) for d in self._dense]).to(dev).contiguous()
        pix, idx, off, conf = cat("pixels", torch.float32), cat("idxs", torch.int32), cat("offsets", torch.float32), \
            cat("confs", torch.float32)
        pts, z = ops.dense_unproject(ctx, start, pix, idx, off, self._res["_core"], self._res["_cam_rows"],
                                     self._res["_base_focals"])
        if clean_depth:
            sizes = torch.tensor([list(np.asarray(im).shape[:2]) for im in self.imgs], dtype=torch.int32, device=dev)
            assert all(int(h) * int(w) == c for (h, w), c in zip(sizes.tolist(), counts)), \
                "clean_depth needs one dense entry per pixel, in raster order"
            conf = ops.dense_clean(ctx, start, sizes, self._res["_cam_rows"], pts, z, conf)
        bounds = start.tolist()
        sl = [slice(bounds[i], bounds[i + 1]) for i in range(len(counts))]
        return [pts[s_] for s_ in sl], list(self.depthmaps), [conf[s_].cpu() for s_ in sl]


def run_sparse_ga(condensed, device="cuda", optim_params=None, lr1=0.07, niter1=500, lr2=0.014, niter2=200, **kw):
    """Global alignment of an already condensed problem (reference run_sparse_ga, reconstruct.py:75-113, from
    the condense_data output onwards) with the reference's schedule (reconstruct.py:61-69)."""
    res, params = align.run(condensed, lr1=lr1, niter1=niter1, lr2=lr2, niter2=niter2, prev_params=optim_params,
                            device=device)
    return SparseGAResult(condensed.get("imgs"), res, condensed.get("dense")), params


def reconstruct_scene(model, imgs, filelist, device, optim_params=None, tmpdir=None):
    """Run the reconstruction pipeline: pairwise inference + matching (`model`), then global alignment.

    Returns (scene, optim_params); pass optim_params back in to warm start after adding images."""
    if tmpdir is None:
        tmpdir = tempfile.mkdtemp()
    if not hasattr(model, "condense"):
        raise NotImplementedError(
            "reconstruct_scene needs a model that implements condense(imgs, filelist, device, cache_dir); the Mast3r "
            "ViT front end (absent from the reference tree: empty submodule) is not bundled -- see the module docstring")
    condensed = model.condense(imgs, filelist, device, tmpdir)
    return run_sparse_ga(condensed, device=device, optim_params=optim_params)


reconstruct = reconstruct_scene  # alias for the wording of BASELINE.json's north_star
