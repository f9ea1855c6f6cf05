"""Opportunistic upstream pins (SURVEY.md 7 "hard part 1").

Paths C and the SSIM loss restate gsplat / torchmetrics, which are absent from /root/reference and from this image, so the
oracle is "parity unpinned vs upstream".  If a box ever carries the real packages these tests compare the HIP path with
THEM (not with the builder's restatement) and the pin comes for free; otherwise they skip.  Tolerances are north_star's:
pixels 1e-4 relative, indices exact.

Reference call sites: starster/gs.py:76-87 (gsplat.rasterization, sh_degree=1, everything else default),
starster/gs.py:39,129 (torchmetrics StructuralSimilarityIndexMeasure(data_range=1))."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from st3r_synth import synth

DEV = "cuda:0"


def _upstream(name):
    """The upstream package `name`, or skip.  ST3R_UPSTREAM_PATH (os.pathsep-separated directories, e.g. a checkout of
    gsplat with its built extension, or a site-packages directory of another environment) is put on sys.path first, so
    a box that carries gsplat / torchmetrics ANYWHERE pins path C without an install (README.md, "Upstream pins")."""
    for d in reversed([d for d in os.environ.get("ST3R_UPSTREAM_PATH", "").split(os.pathsep) if d]):
        if os.path.isdir(d) and d not in sys.path:
            sys.path.insert(0, d)
    try:
        return importlib.import_module(name)
    except Exception as e:   # ImportError, or a binary extension built for another torch / device
        pytest.skip(f"upstream package {name!r} not usable here ({type(e).__name__}: {e}); set ST3R_UPSTREAM_PATH")


def _dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)


@pytest.mark.parametrize("N,V,W,H,lo,hi", [(3000, 2, 128, 96, 0.01, 0.05), (20000, 4, 320, 240, 0.004, 0.03)])
def test_rasterization_against_gsplat(N, V, W, H, lo, hi):
    gsplat = _upstream("gsplat")
    from starst3r_amd import ops
    ctx = ops.get_context(DEV)
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=4, scale_lo=lo, scale_hi=hi)
    P = {k: _dev(v) for k, v in g.items()}
    vm, K = _dev(w2c), _dev(Ks)
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, W, H)
    # the reference's call, argument for argument (starster/gs.py:76-87)
    ref_rgb, ref_alpha, meta = gsplat.rasterization(means=P["means"], quats=P["quats"], scales=P["scales"],
                                                    opacities=P["opacities"], colors=P["shN"], viewmats=vm, Ks=K,
                                                    width=W, height=H, sh_degree=1)
    for key in ("camera_ids", "gaussian_ids", "radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        if key in meta and key in info:
            a, b = info[key].reshape(-1).long().cpu(), meta[key].reshape(-1).long().cpu()
            if key == "radii" and b.numel() == 2 * a.numel():     # newer gsplat: per-axis radii; this build: the 1.4 scalar
                b = b.reshape(-1, 2).max(dim=1).values
            assert torch.equal(a, b), key
    scale = float(ref_rgb.abs().max())
    assert float((rgb - ref_rgb).abs().max()) <= 1e-4 * max(scale, 1.0)
    assert float((alpha - ref_alpha).abs().max()) <= 1e-4
    # gradients through gsplat's autograd vs the fused backward of this build
    Pg = {k: P[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shN")}
    out, _, _ = gsplat.rasterization(means=Pg["means"], quats=Pg["quats"], scales=Pg["scales"], opacities=Pg["opacities"],
                                     colors=Pg["shN"], viewmats=vm, Ks=K, width=W, height=H, sh_degree=1)
    torch.manual_seed(3)
    v_rgb = torch.randn_like(out)
    (out * v_rgb).sum().backward()
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha, info["_last_ids"],
                             v_rgb.contiguous(), None, info["_cum_tiles"], V, W, H)
    grads = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, info["_campos"], W,
                               H, info["_splats"], v_splats)
    G = ops.split_grads(grads, N)
    for k, gk in (("means", "means"), ("quats", "quats"), ("scales", "scales"), ("opacities", "opacities")):
        ref = Pg[k].grad
        assert float((G[gk] - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-9, k
    ref = Pg["shN"].grad[:, :4]
    assert float((G["sh"].reshape(ref.shape) - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-9


@pytest.mark.parametrize("shape", [(1, 96, 128), (3, 240, 320)])
def test_l1_ssim_against_torchmetrics(shape):
    tm = _upstream("torchmetrics")
    from starst3r_amd import ops
    ctx = ops.get_context(DEV)
    V, H, W = shape
    torch.manual_seed(0)
    gt = torch.rand(V, H, W, 3, device=DEV)
    x = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1).requires_grad_(True)
    ssim = tm.StructuralSimilarityIndexMeasure(data_range=1.0).to(DEV)
    # starster/gs.py:126-130, one view at a time
    loss = 0
    for c in range(V):
        l1 = (gt[c] - x[c]).abs().mean()
        s = 1 - ssim(gt[c].permute(2, 0, 1).unsqueeze(0), x[c].permute(2, 0, 1).unsqueeze(0))
        loss = loss + 0.8 * l1 + 0.2 * s
    loss.backward()
    sums, v = ops.loss_l1_ssim(ctx, x.detach().contiguous(), gt, 0.8, 0.2)
    s = sums.cpu().numpy()
    mine = sum(0.8 * s[c, 0] / (H * W * 3) + 0.2 * (1 - s[c, 1] / ((H - 10) * (W - 10) * 3)) for c in range(V))
    assert abs(mine - float(loss)) <= 1e-5 * abs(float(loss))
    assert float((v - x.grad).abs().max()) <= 2e-3 * float(x.grad.abs().max())


def test_mcmc_relocation_against_gsplat():
    gsplat = _upstream("gsplat")
    rel = _upstream("gsplat.relocation")
    from oracle import mcmc_oracle as mo
    rng = np.random.default_rng(0)
    n = 4096
    op = torch.tensor(rng.uniform(0.01, 0.99, n), dtype=torch.float32, device=DEV)
    sc = torch.tensor(rng.uniform(0.001, 0.1, (n, 3)), dtype=torch.float32, device=DEV)
    ratios = torch.tensor(rng.integers(1, 6, n), dtype=torch.int32, device=DEV)
    binoms = torch.tensor(mo.binom_table(), dtype=torch.float32, device=DEV)
    new_op, new_sc = rel.compute_relocation(op, sc, ratios, binoms)
    mo_op, mo_sc = mo.compute_relocation(op.cpu().numpy(), sc.cpu().numpy(), ratios.cpu().numpy())
    np.testing.assert_allclose(new_op.cpu().numpy(), mo_op, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(new_sc.cpu().numpy(), mo_sc, rtol=2e-5, atol=1e-9)
