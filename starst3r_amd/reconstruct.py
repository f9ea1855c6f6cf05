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
        self._dense = dense

    def get_dense_pts3d(self, clean_depth=True):
        """Dense unprojection with the OPTIMISED cameras and depthmaps (Mast3r SparseGA.get_dense_pts3d): every
        dense pixel is treated like an anchor -- depth = depthmap[idx] * offset' -- so the points live in the
        optimiser's gauge (App. A.5 make_pts3d).  `clean_depth` is accepted for API compatibility."""
        if self._dense is None:
            raise NotImplementedError("dense unprojection needs the model's dense pixel table (SURVEY.md 8(f) #3)")
        dev = self.cam2w.device
        pts, confs = [], []
        for i, d in enumerate(self._dense):
            pix = torch.as_tensor(d["pixels"], dtype=torch.float32, device=dev)
            idx = torch.as_tensor(d["idxs"], dtype=torch.int64, device=dev)
            off = torch.as_tensor(d["offsets"], dtype=torch.float32, device=dev)
            K, T = self.intrinsics[i], self.cam2w[i]
            f = K[0, 0]
            offp = 1 + (off - 1) * (float(d["base_focal"]) / f)
            z = self.depthmaps[i][idx] * offp
            pc = torch.stack(((pix[:, 0] - K[0, 2]) / f * z, (pix[:, 1] - K[1, 2]) / f * z, z), dim=-1)
            pts.append(pc @ T[:3, :3].T + T[:3, 3])
            confs.append(torch.as_tensor(d["confs"], dtype=torch.float32))
        return pts, list(self.depthmaps), confs


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
