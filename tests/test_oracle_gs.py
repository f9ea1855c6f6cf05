"""Pins oracle/gs_oracle.c (path C restatement) -- CPU only.

gsplat / torchmetrics are absent ("parity unpinned vs upstream"), so the oracle is pinned
by the KATs SURVEY.md 8(c) lists: (1) single Gaussian, (2) depth-ordered compositing and
last_id, (3) key packing / sort / offsets invariants, (4) fp64 autograd of an independent
dense torch restatement, (5) SSIM identities + autograd, (6) torch.optim.Adam directly.
"""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle as go
from oracle import gs_torch_ref as tr
from st3r_synth import synth
from kat_scenes import cam_front as _cam_front, sh_const as _sh_const, radius_kat_scene as _radius_kat_scene


def test_single_gaussian_kat(oracle_built):
    W, H = 48, 32
    # principal point on the centre of pixel (20,12) and the Gaussian on the optical axis, so the
    # perspective Jacobian has no off-axis term and cov2d is exactly isotropic
    V, K = _cam_front(W, H, cx=20.5, cy=12.5)
    z = 4.0
    means = np.array([[0.0, 0.0, z]], np.float32)
    s = 0.08
    scales = np.full((1, 3), s, np.float32)
    quats = np.array([[1, 0, 0, 0]], np.float32)
    opac = np.array([0.7], np.float32)
    sh = _sh_const(1, (0.2, 0.5, 0.9))
    rgb, alpha, meta = go.rasterization(means, quats, scales, opac, sh, V, K, W, H)
    assert meta["camera_ids"].tolist() == [0] and meta["gaussian_ids"].tolist() == [0]
    np.testing.assert_allclose(meta["means2d"][0], [20.5, 12.5], atol=1e-4)
    # isotropic: cov2d = (f*s/z)^2 + 0.3
    var = (100.0 * s / z) ** 2 + 0.3
    np.testing.assert_allclose(meta["conics"][0], [1 / var, 0, 1 / var], rtol=1e-5, atol=1e-7)
    assert meta["radii"][0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(alpha[0, 12, 20, 0], 0.7, rtol=1e-5)
    np.testing.assert_allclose(rgb[0, 12, 20], 0.7 * np.array([0.2, 0.5, 0.9]), rtol=1e-4)
    # one pixel to the right: alpha = o * exp(-0.5/var)
    np.testing.assert_allclose(alpha[0, 12, 21, 0], 0.7 * math.exp(-0.5 / var), rtol=1e-5)
    # far pixel: below 1/255 -> exactly zero
    assert alpha[0, 0, 47, 0] == 0.0
    # tile rect: radius r around (20.5,12.5) on 16px tiles, 3x2 grid
    r = meta["radii"][0]
    x0, x1 = max(0, math.floor((20.5 - r) / 16)), min(3, math.ceil((20.5 + r) / 16))
    y0, y1 = max(0, math.floor((12.5 - r) / 16)), min(2, math.ceil((12.5 + r) / 16))
    assert meta["tiles_per_gauss"][0] == (x1 - x0) * (y1 - y0)
    # opacity > 0.999 saturates
    rgb2, alpha2, _ = go.rasterization(means, quats, scales, np.array([5.0], np.float32), sh, V, K, W, H)
    np.testing.assert_allclose(alpha2[0, 12, 20, 0], 0.999, rtol=1e-6)


def test_radius_constant_kat(oracle_built):
    """Hand-computed from the formula of SURVEY.md App. A.1, not from the oracle."""
    g, V, K, W, H = _radius_kat_scene()
    _, _, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], V, K, W, H)
    np.testing.assert_allclose(meta["conics"][0], [1 / 0.8, 0, 1 / 0.8], rtol=1e-5, atol=1e-7)
    assert meta["radii"][0] == 3
    # mean at (20.5, 12.5): the square [17.5, 23.5] x [9.5, 15.5] touches tile columns 1 and rows 0 only ... x: floor(17.5/16)=1,
    # ceil(23.5/16)=2 -> one column; y: floor(9.5/16)=0, ceil(15.5/16)=1 -> one row
    assert meta["tiles_per_gauss"][0] == 1


def test_two_gaussians_depth_order_and_last_id(oracle_built):
    W, H = 32, 32
    V, K = _cam_front(W, H, cx=16.5, cy=16.5)
    def mean_at(px, py, z): return [(px - 16.5) * z / 100.0, (py - 16.5) * z / 100.0, z]
    # index 0 is FARTHER than index 1: compositing must put 1 in front
    means = np.array([mean_at(16.5, 16.5, 5.0), mean_at(16.5, 16.5, 3.0)], np.float32)
    scales = np.array([[0.15] * 3, [0.09] * 3], np.float32)
    quats = np.array([[1, 0, 0, 0]] * 2, np.float32)
    opac = np.array([0.6, 0.5], np.float32)
    sh = np.concatenate([_sh_const(1, (1, 0, 0)), _sh_const(1, (0, 1, 0))])
    rgb, alpha, meta = go.rasterization(means, quats, scales, opac, sh, V, K, W, H)
    # sorted flatten ids within each tile: gaussian 1 (near) before 0 (far)
    off = meta["isect_offsets"].reshape(-1); flat = meta["flatten_ids"]
    for t in range(off.size):
        s = off[t]; e = off[t + 1] if t + 1 < off.size else flat.size
        assert list(flat[s:e]) in ([], [1, 0], [1], [0])
    np.testing.assert_allclose(alpha[0, 16, 16, 0], 1 - (1 - 0.5) * (1 - 0.6), rtol=1e-5)
    np.testing.assert_allclose(rgb[0, 16, 16], [0.5 * 0.6, 0.5, 0.0], rtol=1e-4, atol=1e-6)
    # last_id at the centre pixel = sorted position of the far gaussian in its tile
    tile = (16 // 16) * 2 + (16 // 16)
    assert meta["last_ids"][0, 16, 16] == off[tile] + 1
    assert flat[meta["last_ids"][0, 16, 16]] == 0


def test_keys_sort_offsets_invariants(oracle_built):
    g, w2c, Ks = synth.make_scene(400, 3, 96, 64, seed=7, scale_lo=0.01, scale_hi=0.08)
    rgb, alpha, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, 96, 64)
    tw, th, Cn = meta["tile_width"], meta["tile_height"], 3
    assert (tw, th) == (6, 4)
    tile_bits = (tw * th).bit_length()
    ids = meta["isect_ids"]
    assert ids.size == meta["tiles_per_gauss"].sum() > 0
    assert np.all(np.diff(ids) >= 0)
    cam = ids >> (32 + tile_bits); tile = (ids >> 32) & ((1 << tile_bits) - 1)
    depth = (ids & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    flat = meta["flatten_ids"]
    assert np.array_equal(cam, meta["camera_ids"][flat])
    assert np.array_equal(depth, meta["depths"][flat])
    assert tile.max() < tw * th
    # stable: equal keys keep ascending packed index
    same = np.diff(ids) == 0
    assert np.all(np.diff(flat)[same] > 0)
    # offsets = searchsorted of (cam,tile) ids
    ct = cam * (tw * th) + tile
    expect = np.searchsorted(ct, np.arange(Cn * tw * th), side="left")
    assert np.array_equal(meta["isect_offsets"].reshape(-1), expect)
    # unsorted emission is y-major then x within each gaussian, cumulative
    cum = np.concatenate([[0], np.cumsum(meta["tiles_per_gauss"])])
    un = meta["isect_ids_unsorted"]; unf = meta["flatten_ids_unsorted"]
    for i in np.nonzero(meta["tiles_per_gauss"] > 1)[0][:20]:
        seg = (un[cum[i]:cum[i + 1]] >> 32) & ((1 << tile_bits) - 1)
        assert np.all(np.diff(seg) > 0) and np.all(unf[cum[i]:cum[i + 1]] == i)
    # every pixel's alpha in [0,1)
    assert alpha.min() >= 0 and alpha.max() < 1.0


def _small_scene(seed, n=60, W=40, H=24, views=2):
    g, w2c, Ks = synth.make_scene(n, views, W, H, seed=seed, scale_lo=0.03, scale_hi=0.25, extent=0.9)
    return g, w2c, Ks, W, H


@pytest.mark.parametrize("seed", [3, 11])
def test_forward_matches_dense_torch(oracle_built, seed):
    g, w2c, Ks, W, H = _small_scene(seed)
    rgb, alpha, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H,
                                        want_margin=True)
    N, Cn = g["means"].shape[0], w2c.shape[0]
    vis = np.zeros((Cn, N), bool); rad = np.zeros((Cn, N), np.int64)
    vis[meta["camera_ids"], meta["gaussian_ids"]] = True
    rad[meta["camera_ids"], meta["gaussian_ids"]] = meta["radii"]
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    rgb_t, alpha_t = tr.render_dense(t(g["means"]), t(g["quats"]), t(g["scales"]), t(g["opacities"]), t(g["shN"]),
                                     t(w2c), t(Ks), W, H, torch.tensor(vis), torch.tensor(rad))
    ok = meta["margin"] > 1e-4  # pixels whose skip/stop decisions are not borderline
    assert ok.mean() > 0.99
    np.testing.assert_allclose(rgb[ok], rgb_t.numpy()[ok], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(alpha[ok], alpha_t.numpy()[ok], rtol=2e-4, atol=2e-5)
    assert alpha.max() > 0.5  # the scene actually renders something


@pytest.mark.parametrize("seed", [3, 11])
def test_backward_matches_fp64_autograd(oracle_built, seed):
    g, w2c, Ks, W, H = _small_scene(seed)
    rgb, alpha, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
    rng = np.random.default_rng(seed)
    v_rgb = rng.standard_normal(rgb.shape).astype(np.float32)
    v_alpha = rng.standard_normal(alpha.shape).astype(np.float32)
    grads = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H,
                                      meta, alpha, v_rgb, v_alpha)
    N, Cn = g["means"].shape[0], w2c.shape[0]
    vis = np.zeros((Cn, N), bool); rad = np.zeros((Cn, N), np.int64)
    vis[meta["camera_ids"], meta["gaussian_ids"]] = True
    rad[meta["camera_ids"], meta["gaussian_ids"]] = meta["radii"]
    P = {k: torch.tensor(g[k], dtype=torch.float64, requires_grad=True) for k in ("means", "quats", "scales", "opacities", "shN")}
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    rgb_t, alpha_t = tr.render_dense(P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], t(w2c), t(Ks), W, H,
                                     torch.tensor(vis), torch.tensor(rad))
    (rgb_t * t(v_rgb)).sum().add((alpha_t * t(v_alpha)).sum()).backward()
    def close(a, b, name):
        a = np.asarray(a); b = b.numpy()
        scale = np.abs(b).max() + 1e-12
        err = np.abs(a - b).max() / scale
        assert err < 2e-3, (name, err)
    close(grads["means"], P["means"].grad, "means")
    close(grads["quats"], P["quats"].grad, "quats")
    close(grads["scales"], P["scales"].grad, "scales")
    close(grads["opacities"], P["opacities"].grad, "opacities")
    close(grads["sh"], P["shN"].grad[:, :4], "sh")
    assert float(P["shN"].grad[:, 4:].abs().max()) == 0.0  # rows 4..23 never receive gradient


def test_ssim_identities_and_autograd(oracle_built):
    rng = np.random.default_rng(0)
    H, W = 30, 37
    x = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    l1, ss, _ = go.l1_ssim(x, x, want_grad=False)
    assert l1 == 0.0 and abs(ss - 1.0) < 1e-12
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    l1, ss, vr = go.l1_ssim(x, y, 0.8, 0.2)
    X = torch.tensor(x, dtype=torch.float64, requires_grad=True); Y = torch.tensor(y, dtype=torch.float64)
    l1_t = (Y - X).abs().mean(); ss_t = tr.ssim_mean(Y, X)
    (0.8 * l1_t + 0.2 * (1 - ss_t)).backward()
    assert abs(l1 - float(l1_t)) < 1e-7 and abs(ss - float(ss_t)) < 1e-7
    np.testing.assert_allclose(vr, X.grad.numpy(), rtol=1e-4, atol=1e-9)
    # constant images: ssim = (2ab + c1)/(a^2+b^2+c1) (variances vanish)
    a, b = 0.3, 0.6
    _, ssc, _ = go.l1_ssim(np.full((24, 24, 3), a, np.float32), np.full((24, 24, 3), b, np.float32), want_grad=False)
    c1 = 1e-4
    assert abs(ssc - (2 * a * b + c1) / (a * a + b * b + c1)) < 1e-5


def test_adam_matches_torch_optim(oracle_built):
    rng = np.random.default_rng(5)
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    P = torch.nn.Parameter(torch.tensor(p0.copy()))
    opt = torch.optim.Adam([P], lr=1e-3)
    p = p0.copy(); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    for step in range(1, 8):
        gnp = (rng.standard_normal(n) * (10.0 ** rng.integers(-4, 2, n))).astype(np.float32)
        P.grad = torch.tensor(gnp.copy()); opt.step()
        go.adam(p, gnp, m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        np.testing.assert_allclose(p, P.detach().numpy(), rtol=0, atol=2e-7)
    st = opt.state[P]
    np.testing.assert_allclose(m, st["exp_avg"].numpy(), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(v, st["exp_avg_sq"].numpy(), rtol=1e-6, atol=1e-20)


def test_ssim_against_scipy_gaussian_filter(oracle_built):
    """An implementation of the windowed moments that is NOT the builder's: scipy.ndimage.gaussian_filter with
    sigma 1.5, truncate 3.5 (= the 11-tap window of Wang et al. 2004, which torchmetrics' data_range=1 default and
    scikit-image's gaussian_weights=True both use), population statistics, K1 = .01, K2 = .03, mean over the pixels whose
    window lies inside the image (what torchmetrics keeps after cropping its padding; starster/gs.py:39,129)."""
    nd = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(7)
    H, W = 41, 53
    x = rng.uniform(0, 1, (H, W, 3))
    y = np.clip(x + rng.normal(0, 0.15, x.shape), 0, 1)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    vals = []
    for ch in range(3):
        f = lambda a: nd.gaussian_filter(a, 1.5, truncate=3.5, mode="constant")[5:-5, 5:-5]
        mx, my = f(x[..., ch]), f(y[..., ch])
        sxx, syy, sxy = f(x[..., ch] ** 2) - mx * mx, f(y[..., ch] ** 2) - my * my, f(x[..., ch] * y[..., ch]) - mx * my
        vals.append(((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2)))
    want = float(np.mean(vals))
    _, ss, _ = go.l1_ssim(x.astype(np.float32), y.astype(np.float32), want_grad=False)
    assert abs(ss - want) < 2e-6, (ss, want)
