#!/usr/bin/env python
"""Generates tests/golden/gs_loop_*.npz by RUNNING the reference's own 3DGS glue
(/root/reference/starster/gs.py: `init_3dgs` :14-45, `render_3dgs` :47-88, `render_3dgs_original` :90-95,
`run_3dgs_optim` :97-166 with its inner `compute_loss` :126-136) in this container.

The reference file is executed from where it lies (never copied).  Its two absent third-party imports are
registered as stub modules before it is loaded:

  gsplat.rasterization              -> a torch.autograd.Function over oracle/gs_oracle.c's forward and analytic backward
                                       (the restatement of SURVEY App. A.1; float32 forward, float64 backward sums)
  gsplat.MCMCStrategy               -> a RECORDER: check_sanity / initialize_state / step_pre_backward /
                                       step_post_backward log their arguments and change nothing
  torchmetrics.image.StructuralSimilarityIndexMeasure
                                    -> an nn.Module over oracle/gs_torch_ref.ssim_mean (torch, differentiable; App. A.3)

What these vectors PIN (it is executed, not read): the parameter dictionary of `init_3dgs` (raw scales / opacities, wxyz,
1 - colour in sh0 and in all 24 shN rows), the 0.8 / 0.2 loss mix, both regularisers once PER VIEW through autograd, the
six `torch.optim.Adam(lr)` with `sh0` never receiving a gradient and shN rows 4..23 receiving zeros, the per-iteration
`loss.item()` list, the hook sequence with `step` restarting per call and the literal 1e-3.
What they do NOT pin: gsplat's / torchmetrics' own arithmetic -- the stubs are this repository's restatement [U].

Only numeric arrays are written.  Run:  python tools/gen_gs_goldens.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/starster/gs.py"

from oracle import gs_oracle as go          # noqa: E402
from oracle import gs_torch_ref as gt_ref   # noqa: E402

KEYS = ("means", "scales", "quats", "opacities", "sh0", "shN")


# ---------------------------------------------------------------- stubs for the absent packages
class _Raster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmats, Ks, width, height):
        a = [t.detach().cpu().numpy() for t in (means, quats, scales, opacities, colors, viewmats, Ks)]
        rgb, alpha, meta = go.rasterization(*a, width, height, sh_degree=1)
        ctx.a, ctx.meta, ctx.alpha, ctx.wh = a, meta, alpha, (width, height)
        return torch.from_numpy(rgb), torch.from_numpy(alpha)

    @staticmethod
    def backward(ctx, v_rgb, v_alpha):
        a, (W, H) = ctx.a, ctx.wh
        va = None if v_alpha is None or not bool(v_alpha.abs().sum() > 0) else v_alpha.numpy()
        G = go.rasterization_backward(*a, W, H, ctx.meta, ctx.alpha, v_rgb.contiguous().numpy(), va)
        N = a[0].shape[0]
        v_sh = np.zeros(a[4].shape, np.float32)           # rows 4..23: zeros, not None (autograd of an indexed read)
        v_sh[:, :4] = np.asarray(G["sh"], np.float32).reshape(N, 4, 3)
        f = lambda k, shp: torch.from_numpy(np.asarray(G[k], np.float32).reshape(shp))
        return (f("means", (N, 3)), f("quats", (N, 4)), f("scales", (N, 3)), f("opacities", (N,)),
                torch.from_numpy(v_sh), None, None, None, None)


RASTER_CALLS = []


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree=None, **kw):
    assert not kw, f"the reference passes no further arguments (gs.py:76-87): {kw}"
    RASTER_CALLS.append(dict(sh_degree=sh_degree, colors_shape=tuple(colors.shape), width=width, height=height))
    rgb, alpha = _Raster.apply(means, quats, scales, opacities, colors, viewmats, Ks, width, height)
    info = {"width": width, "height": height, "n_cameras": viewmats.shape[0]}
    return rgb, alpha, info


class MCMCStrategy:
    """Recorder standing where gsplat.MCMCStrategy() is constructed (gs.py:43)."""
    log = []

    def __init__(self, *a, **kw):
        MCMCStrategy.log.append(("init", len(a), tuple(sorted(kw))))

    def check_sanity(self, params, optimizers):
        MCMCStrategy.log.append(("check_sanity", tuple(params.keys()), tuple(optimizers.keys())))
        assert set(params) == set(optimizers)

    def initialize_state(self, *a, **kw):
        MCMCStrategy.log.append(("initialize_state", len(a), tuple(sorted(kw))))
        return {"binoms": None}

    def step_pre_backward(self, params, optimizers, state, step, info):
        MCMCStrategy.log.append(("pre", int(step), float("nan"), int(len(RASTER_CALLS))))

    def step_post_backward(self, params, optimizers, state, step, info, lr):
        # by now optim.step() and zero_grad(set_to_none=True) of the same iteration have run (gs.py:159-164)
        grads_none = all(p.grad is None for p in params.values())
        MCMCStrategy.log.append(("post", int(step), float(lr), int(grads_none)))


class StructuralSimilarityIndexMeasure(torch.nn.Module):
    def __init__(self, data_range=None, **kw):
        super().__init__()
        assert data_range == 1 and not kw, "the reference constructs SSIM(data_range=1) (gs.py:39)"

    def forward(self, preds, target):
        assert preds.shape[0] == 1 and preds.shape[1] == 3 and preds.shape == target.shape
        return gt_ref.ssim_mean(preds[0].permute(1, 2, 0), target[0].permute(1, 2, 0))


def load_reference_gs():
    gsplat = types.ModuleType("gsplat")
    gsplat.rasterization = rasterization
    gsplat.MCMCStrategy = MCMCStrategy
    tm = types.ModuleType("torchmetrics"); tmi = types.ModuleType("torchmetrics.image")
    tmi.StructuralSimilarityIndexMeasure = StructuralSimilarityIndexMeasure
    tm.image = tmi
    sys.modules.update({"gsplat": gsplat, "torchmetrics": tm, "torchmetrics.image": tmi})
    spec = importlib.util.spec_from_file_location("reference_starster_gs", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------- a scene object with what gs.py touches
class FakeScene:
    """The attributes and methods of starster.Scene that starster/gs.py uses (scene.py:79-95, 157-183)."""

    def __init__(self, ref, pts, cols, imgs, c2w, Ks):
        self._ref = ref
        self.device = "cpu"
        self.dense_pts_flat = torch.tensor(pts)
        self.dense_cols_flat = torch.tensor(cols)
        self.imgs = imgs
        self.c2w = torch.tensor(c2w)
        self.intrinsics = torch.tensor(Ks)

    @property
    def w2c(self):                         # scene.py:91-95
        return torch.inverse(self.c2w)

    def render_3dgs(self, w2c, intrinsics, width, height):   # scene.py:160
        return self._ref.render_3dgs(self, w2c, intrinsics, width, height)

    def render_3dgs_original(self, width, height):           # scene.py:163
        return self._ref.render_3dgs_original(self, width, height)


def look_at(eye, target=(0, 0, 0)):
    eye = np.asarray(eye, np.float64); f = np.asarray(target, np.float64) - eye; f /= np.linalg.norm(f)
    r = np.cross(f, [0, 0, 1.0]); r /= np.linalg.norm(r); d = np.cross(f, r)
    c2w = np.eye(4); c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, d, f, eye
    return c2w.astype(np.float32)


def make_inputs(seed, n_pts, n_views, W, H, gt_scale):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-0.5, 0.5, (n_pts, 3)).astype(np.float32)
    cols = rng.uniform(0.05, 0.95, (n_pts, 3)).astype(np.float32)
    c2w = np.stack([look_at((2.2 * np.cos(a), 2.2 * np.sin(a), 0.5 + 0.2 * k))
                    for k, a in enumerate(np.linspace(0.3, 2.4, n_views))])
    fx = 0.5 * W / np.tan(np.radians(28.0))
    Ks = np.tile(np.array([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], np.float32), (n_views, 1, 1))
    # ground-truth photographs: the same points, blobs of size gt_scale, shifted and recoloured (loss != 0)
    gm = pts + rng.normal(0, 0.01, pts.shape).astype(np.float32)
    gq = np.zeros((n_pts, 4), np.float32); gq[:, 0] = 1
    gs = np.full((n_pts, 3), gt_scale, np.float32)
    gsh = np.zeros((n_pts, 24, 3), np.float32); gsh[:, 0] = (cols - 0.5) / 0.2820947917738781
    w2c = np.linalg.inv(c2w.astype(np.float64)).astype(np.float32)
    rgb, _, _ = go.rasterization(gm, gq, gs, np.full(n_pts, 0.8, np.float32), gsh, w2c, Ks, W, H)
    imgs = [np.clip(rgb[i], 0, 1).astype(np.float32) for i in range(n_views)]
    return pts, cols, imgs, c2w, Ks


def snapshot(scene, tag, out):
    for k in KEYS:
        out[f"{tag}_{k}"] = scene.gaussians[k].detach().numpy().copy()
    for k in KEYS:
        st = scene.optimizers[k].state
        has = len(st) > 0
        out[f"{tag}_adam_has_state_{k}"] = np.array(int(has))
        if has:
            s = next(iter(st.values()))
            out[f"{tag}_adam_step_{k}"] = np.array(float(s["step"]))
            out[f"{tag}_adam_m_{k}"] = s["exp_avg"].numpy().copy()
            out[f"{tag}_adam_v_{k}"] = s["exp_avg_sq"].numpy().copy()


def run_case(ref, name, seed, n_pts, n_views, W, H, gt_scale, init_kw, optim_kw, segments, pruning):
    """segments: iteration counts of successive run_3dgs_optim calls (step restarts at 0 in each, gs.py:143)."""
    torch.manual_seed(0)
    MCMCStrategy.log.clear(); RASTER_CALLS.clear()
    pts, cols, imgs, c2w, Ks = make_inputs(seed, n_pts, n_views, W, H, gt_scale)
    scene = FakeScene(ref, pts, cols, imgs, c2w, Ks)
    ref.init_3dgs(scene, **init_kw)
    out = dict(pts=pts, cols=cols, imgs=np.stack(imgs), c2w=c2w, Ks=Ks, W=np.array(W), H=np.array(H),
               segments=np.array(segments), pruning=np.array(int(pruning)),
               init_scale=np.array(init_kw.get("init_scale", 3e-3)), lr=np.array(init_kw.get("lr", 1e-3)),
               loss_ssim_fac=np.array(optim_kw.get("loss_ssim_fac", 0.2)),
               loss_opacity_fac=np.array(optim_kw.get("loss_opacity_fac", 0.01)),
               loss_scale_fac=np.array(optim_kw.get("loss_scale_fac", 0.01)))
    out["init_is_parameter"] = np.array([int(isinstance(scene.gaussians[k], torch.nn.Parameter)) for k in KEYS])
    out["init_lr_of_optimizers"] = np.array([scene.optimizers[k].param_groups[0]["lr"] for k in KEYS])
    snapshot(scene, "init", out)
    # the render the first iteration sees (reference's render_3dgs_original through the stub)
    with torch.no_grad():
        r0, a0, _ = scene.render_3dgs_original(W, H)
    out["render0"] = r0.numpy().copy(); out["alpha0"] = a0.numpy().copy()
    RASTER_CALLS.clear()
    losses, total = [], 0
    for n in segments:
        ret = ref.run_3dgs_optim(scene, n, enable_pruning=pruning, **optim_kw)
        assert isinstance(ret, list) and len(ret) == n and all(isinstance(x, float) for x in ret)
        losses += ret; total += n
        snapshot(scene, f"it{total}", out)
    out["losses"] = np.array(losses, np.float64)
    out["raster_sh_degree"] = np.array([c["sh_degree"] for c in RASTER_CALLS])
    out["raster_colors_rows"] = np.array([c["colors_shape"][1] for c in RASTER_CALLS])
    hook = [e for e in MCMCStrategy.log if e[0] in ("pre", "post")]
    out["hook_kind"] = np.array([0 if e[0] == "pre" else 1 for e in hook], np.int32)
    out["hook_step"] = np.array([e[1] for e in hook], np.int32)
    out["hook_lr"] = np.array([e[2] for e in hook], np.float64)
    out["hook_aux"] = np.array([e[3] for e in hook], np.int32)   # pre: rasterizations so far; post: all grads None
    setup = [e for e in MCMCStrategy.log if e[0] not in ("pre", "post")]
    out["setup_calls"] = np.array([("init", "check_sanity", "initialize_state").index(e[0]) for e in setup], np.int32)
    path = os.path.join(ROOT, "tests", "golden", f"gs_loop_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: N={n_pts} views={n_views} {W}x{H} segments={segments} pruning={pruning} "
          f"loss {losses[0]:.6f} -> {losses[-1]:.6f}  ({os.path.getsize(path) / 1024:.0f} KB)")


def main():
    ref = load_reference_gs()
    # the reference's defaults (init_scale 3e-3, lr 1e-3, facs 0.2 / 0.01 / 0.01), snapshots after 1, 5 and 20 iterations
    run_case(ref, "default", 11, 300, 3, 64, 48, 0.03, {}, {}, [1, 4, 15], False)
    # the same through the strategy hooks (recorder): call order, step restarting per call, the literal 1e-3
    run_case(ref, "hooks", 11, 300, 3, 64, 48, 0.03, {}, {}, [1, 4, 15], True)
    # non-default arguments: larger blobs that overlap and saturate, another lr, other loss factors, two views
    run_case(ref, "args", 12, 260, 2, 80, 48, 0.05, dict(init_scale=0.04, lr=2e-3),
             dict(loss_ssim_fac=0.35, loss_opacity_fac=0.02, loss_scale_fac=0.05), [2, 8], False)


if __name__ == "__main__":
    main()
