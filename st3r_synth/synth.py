"""Synthetic scenes for benchmarks and parity tests (SURVEY.md 8(d) `SYNTH-1M`).

Pure numpy, deterministic (PCG64).  Layouts follow the reference's own containers:
gaussians dict keys/shapes of starster/gs.py:20-27 (scales/opacities RAW, quats wxyz),
cameras as world-to-camera 4x4 (starster/scene.py:91-95) + 3x3 intrinsics, OpenCV axes.
"""
import math

import numpy as np

SEED = 20241220


def look_at_w2c(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """OpenCV camera (x right, y down, z forward) looking from `eye` to `target`."""
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64); up = np.asarray(up, np.float64)
    f = target - eye; f /= np.linalg.norm(f)
    r = np.cross(f, up); r /= np.linalg.norm(r)
    d = np.cross(f, r)  # "down"
    R = np.stack([r, d, f])  # rows: camera axes in world coords
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = -R @ eye
    return M


def make_cameras(n_views, width, height, radius=3.5, cam_height=0.8, hfov_deg=60.0):
    fx = 0.5 * width / math.tan(math.radians(hfov_deg) / 2)
    Ks = np.zeros((n_views, 3, 3), np.float32)
    w2c = np.zeros((n_views, 4, 4), np.float32)
    for k in range(n_views):
        az = 2 * math.pi * k / max(n_views, 8) if n_views <= 8 else 2 * math.pi * k / n_views
        eye = (radius * math.cos(az), radius * math.sin(az), cam_height)
        w2c[k] = look_at_w2c(eye).astype(np.float32)
        Ks[k] = np.array([[fx, 0, width / 2], [0, fx, height / 2], [0, 0, 1]], np.float32)
    return w2c, Ks


def make_gaussians(n, seed=SEED, scale_lo=0.002, scale_hi=0.008, extent=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    means = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    scales = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (n, 3))).astype(np.float32)
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 1.0, n).astype(np.float32)
    shN = np.zeros((n, 24, 3), np.float32)
    shN[:, 0:4, :] = rng.uniform(-0.5, 0.5, (n, 4, 3)).astype(np.float32)
    sh0 = np.zeros((n, 1, 3), np.float32)
    return dict(means=means, scales=scales, quats=q.astype(np.float32), opacities=opac, sh0=sh0, shN=shN)


def perturb_for_gt(g, seed=SEED + 1, sigma=0.002):
    """Ground-truth scene = same scene with jittered means and re-drawn SH (so loss != 0)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    gt = {k: v.copy() for k, v in g.items()}
    gt["means"] = (g["means"] + rng.normal(0, sigma, g["means"].shape)).astype(np.float32)
    gt["shN"][:, 0:4, :] = rng.uniform(-0.5, 0.5, (g["means"].shape[0], 4, 3)).astype(np.float32)
    return gt


def make_scene(n, n_views, width, height, seed=SEED, **kw):
    g = make_gaussians(n, seed, **{k: v for k, v in kw.items() if k in ("scale_lo", "scale_hi", "extent")})
    w2c, Ks = make_cameras(n_views, width, height, **{k: v for k, v in kw.items() if k in ("radius", "cam_height", "hfov_deg")})
    return g, w2c, Ks
