"""Hand-built known-answer scenes shared by the CPU oracle tests and the GPU parity tests."""
import math

import numpy as np


def cam_front(W, H, f=100.0, cx=None, cy=None):
    """camera at origin looking down +z (identity viewmat)."""
    V = np.eye(4, dtype=np.float32)[None]
    cx = W / 2 if cx is None else cx; cy = H / 2 if cy is None else cy
    K = np.array([[[f, 0, cx], [0, f, cy], [0, 0, 1]]], np.float32)
    return V, K


def sh_const(n, rgb):
    """SH coefficients giving colour `rgb` for every direction (only k0 non-zero)."""
    sh = np.zeros((n, 24, 3), np.float32)
    sh[:, 0, :] = (np.asarray(rgb, np.float32) - 0.5) / 0.2820947917738781
    return sh


def radius_kat_scene():
    """One isotropic Gaussian on the optical axis whose radius separates gsplat's max(0.01, b^2 - det) from the
    INRIA rasterizer's 0.1: cov2d = var * I with var = (f s / z)^2 + 0.3 = 0.8, so b^2 - det = 0 and
    radius = ceil(3 sqrt(var + sqrt(0.01))) = ceil(3 sqrt(0.9)) = ceil(2.846) = 3   (SURVEY.md App. A.1);
    the 0.1 constant would give ceil(3 sqrt(0.8 + 0.3162)) = ceil(3.169) = 4."""
    W, H = 48, 32
    V, K = cam_front(W, H, cx=20.5, cy=12.5)
    z = 4.0
    s = math.sqrt(0.5) * z / 100.0
    return dict(means=np.array([[0.0, 0.0, z]], np.float32), scales=np.full((1, 3), s, np.float32),
                quats=np.array([[1, 0, 0, 0]], np.float32), opacities=np.array([0.7], np.float32),
                shN=sh_const(1, (0.2, 0.5, 0.9))), V, K, W, H

