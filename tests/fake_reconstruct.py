"""A deterministic stand-in for `reconstruct_scene` (starster/reconstruct.py:19-72) -- TEST INFRASTRUCTURE shared by
tools/gen_scene_goldens.py (which drives the REFERENCE's Scene with it) and tests/test_host_scene_golden.py (which drives this
repository's Scene with it): the object it returns has exactly the members starster/scene.py:122-155 touches
(`imgs`, `cam2w`, `intrinsics`, `get_dense_pts3d(clean_depth=True)`)."""
import numpy as np
import torch

H, W = 6, 8


class FakeSparseGA:
    def __init__(self, n):
        rng = np.random.default_rng(1000 + n)
        self.imgs = [rng.uniform(0, 1, (H, W, 3)).astype(np.float32) for _ in range(n)]
        self.cam2w = torch.tensor(rng.normal(size=(n, 4, 4)).astype(np.float32)) + 3 * torch.eye(4)
        self.intrinsics = torch.tensor(rng.uniform(1, 2, (n, 3, 3)).astype(np.float32))
        self._pts = [torch.tensor(rng.normal(size=(H * W, 3)).astype(np.float32)) for _ in range(n)]
        self._confs = [torch.tensor(rng.uniform(0.5, 3.0, (H, W)).astype(np.float32)) for _ in range(n)]
        self.dense_calls = []

    def get_dense_pts3d(self, clean_depth=False, **kw):
        self.dense_calls.append((bool(clean_depth), tuple(sorted(kw))))
        return self._pts, None, self._confs


class Recorder:
    """reconstruct_scene(model, imgs, filelist, device, optim_params=None, tmpdir=None) -> (scene, optim_params)"""

    def __init__(self):
        self.calls, self.results = [], []

    def __call__(self, model, imgs, filelist, device, optim_params=None, tmpdir=None):
        token = {"call": len(self.calls)}                       # what the reference hands back in as the warm start
        self.calls.append(dict(model=model, n_imgs=len(imgs), filelist=list(filelist), device=str(device),
                               optim_params_in=None if optim_params is None else optim_params["call"], tmpdir=tmpdir))
        res = FakeSparseGA(len(imgs))
        self.results.append(res)
        return res, token


def raw_images(k0, k1):
    return [torch.full((5, 7, 3), float(i)) for i in range(k0, k1)]
