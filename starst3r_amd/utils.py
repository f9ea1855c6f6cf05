"""SE(3) interpolation helpers (public API of the reference: starster/utils.py:13-78,
docs/api.rst "Utils").  Host-side torch math, not part of the hot path."""
import torch

__all__ = ("interp_se3", "interp_se3_path")


def interp_se3(mat1: torch.Tensor, mat2: torch.Tensor, fac: float) -> torch.Tensor:
    """Blend two (4,4) rigid transforms: translation and rotation entries are interpolated
    linearly, then the rotation columns are re-orthonormalised (Gram-Schmidt, column 0 first)."""
    out = torch.zeros_like(mat1)
    out[3, 3] = 1
    out[:3, 3] = mat1[:3, 3] + (mat2[:3, 3] - mat1[:3, 3]) * fac
    rot = mat1[:3, :3] + (mat2[:3, :3] - mat1[:3, :3]) * fac
    c0, c1, c2 = rot[:, 0].clone(), rot[:, 1].clone(), rot[:, 2].clone()
    c1 = c1 - c0 * c0.dot(c1)
    c2 = c2 - c0 * c0.dot(c2)
    c2 = c2 - c1 * c1.dot(c2)
    basis = torch.stack((c0, c1, c2), dim=1)
    out[:3, :3] = basis / torch.linalg.norm(basis, dim=0)
    return out


def interp_se3_path(mat1: torch.Tensor, mat2: torch.Tensor, steps: int) -> torch.Tensor:
    """(steps,4,4) stack of interp_se3 at evenly spaced factors from 0 to 1."""
    return torch.stack([interp_se3(mat1, mat2, f) for f in torch.linspace(0, 1, steps)], dim=0)
