"""Scene -- drop-in for the reference's `starster.Scene` (starster/scene.py:19-183): same constructor,
attributes and methods.  State holder only; the work happens in reconstruct.py (paths A/B) and gs.py (path C).
"""
__all__ = ("Scene",)

import tempfile
from typing import Any, Optional

import torch

from . import gs as _gs
from .reconstruct import reconstruct_scene


class Scene:
    """State of one reconstruction: the input photographs, the poses / intrinsics / dense points the alignment produced for
    them, and -- after init_3dgs -- the Gaussians with their optimisers.  Mirrors starster/scene.py:19-77 attribute for
    attribute (the boundary users program against); every method hands the work to reconstruct.py or gs.py."""

    def __init__(self, cache_dir: Optional[str] = None, device="cuda"):
        self.device = device
        self.cache_dir = cache_dir if cache_dir is not None else tempfile.mkdtemp()
        self.raw_imgs = []
        self.imgs = []
        self.dense_pts = []
        self.dense_cols = []
        self.c2w = None
        self.intrinsics = None
        self.optim_params = None
        # declared by the reference but never written (scene.py:42-45,74-77); the names in use are the ones
        # init_3dgs sets: gaussians, optimizers, ssim, strategy, strategy_state (gs.py:20,37,39,43,45)
        self.gs_params = None
        self.gs_optims = None
        self.gs_strategy = None
        self.gs_state = None
        self._w2c_cache = None

    @property
    def dense_pts_flat(self):
        """One [sum_i n_i, 3] tensor: the per-view point sets of `dense_pts` stacked in view order (scene.py:79-83)."""
        assert self.dense_pts, "No dense points available."
        return torch.cat(self.dense_pts, dim=0)

    @property
    def dense_cols_flat(self):
        """The colours that go with `dense_pts_flat`, same order (scene.py:85-89)."""
        assert self.dense_cols, "No dense colors available."
        return torch.cat(self.dense_cols, dim=0)

    @property
    def w2c(self) -> torch.Tensor:
        """World-to-camera transformation matrix (inverse of ``c2w``); the reference re-inverts on every
        access (scene.py:91-95, i.e. every training iteration) -- here the inverse is cached per c2w tensor."""
        assert self.c2w is not None, "No c2w matrix available."
        c = self._w2c_cache
        if c is None or c[0] is not self.c2w or c[1] != self.c2w._version:
            self._w2c_cache = (self.c2w, self.c2w._version, torch.inverse(self.c2w))
        return self._w2c_cache[2]

    def add_images(self, model, imgs, conf_thres=1.5):
        """Append photographs (each [H, W, 3]) and solve the WHOLE set again: matching, alignment (warm-started from the
        previous call's parameters) and dense points, keeping per view the points whose confidence exceeds `conf_thres`.
        Poses, intrinsics and points of earlier views are replaced, not extended (scene.py:97-155); the
        pair cache under `cache_dir` is keyed by the stand-in file names "0.png", "1.png", ... like the reference's."""
        self.raw_imgs.extend(imgs)
        names = [f"{i}.png" for i in range(len(self.raw_imgs))]
        result, self.optim_params = reconstruct_scene(model, self.raw_imgs, names, self.device,
                                                      optim_params=self.optim_params, tmpdir=self.cache_dir)
        self.imgs.extend(result.imgs[len(self.imgs):])
        self.c2w, self.intrinsics = result.cam2w, result.intrinsics
        pts, _, confs = result.get_dense_pts3d(clean_depth=True)
        keep = [(confs[i] > conf_thres).reshape(-1).cpu() for i in range(len(result.imgs))]
        self.dense_pts = [pts[i].cpu()[k] for i, k in enumerate(keep)]
        self.dense_cols = [torch.as_tensor(result.imgs[i]).reshape(-1, 3)[k] for i, k in enumerate(keep)]

    def init_3dgs(self, init_scale=3e-3, lr=1e-3):
        _gs.init_3dgs(self, init_scale, lr)

    def render_3dgs(self, w2c, intrinsics, width, height):
        return _gs.render_3dgs(self, w2c, intrinsics, width, height)

    def render_3dgs_original(self, width, height):
        return _gs.render_3dgs_original(self, width, height)

    def run_3dgs_optim(self, iters: int, enable_pruning: bool = False, loss_ssim_fac=0.2, loss_opacity_fac=0.01,
                       loss_scale_fac=0.01, verbose: bool = False) -> list:
        return _gs.run_3dgs_optim(self, iters, enable_pruning, loss_ssim_fac, loss_opacity_fac, loss_scale_fac,
                                  verbose)
