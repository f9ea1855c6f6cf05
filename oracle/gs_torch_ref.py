"""Dense, differentiable torch restatement of the rasterization (TEST INFRASTRUCTURE ONLY).

Independent of gs_oracle.c: no tiles-lists, no sort-by-key, no hand-written backward.
Every (pixel, gaussian) pair is evaluated densely in float64 and torch autograd supplies
the gradients.  Used to pin gs_oracle.c's analytic backward (KAT (4) of SURVEY.md 8(c)).
The only thing borrowed from the C oracle is the *non-differentiable* visibility/tile
rectangle (radii), exactly as gsplat treats it [U].
"""
import math

import torch

SH_C0 = 0.2820947917738781
SH_C1 = 0.48860251190292


def quat_to_rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmat, K, W, H, eps2d=0.3):
    """-> mean2d [N,2], depth [N], conic [N,3] (a,b,c), cov2d for one camera (all differentiable)."""
    R = viewmat[:3, :3]; t = viewmat[:3, 3]
    pc = means @ R.T + t
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    cov = M @ M.transpose(1, 2)
    covc = R @ cov @ R.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = pc.unbind(-1)
    tan_fovx = 0.5 * W / fx; tan_fovy = 0.5 * H / fy
    lim_xp = (W - cx) / fx + 0.3 * tan_fovx; lim_xn = cx / fx + 0.3 * tan_fovx
    lim_yp = (H - cy) / fy + 0.3 * tan_fovy; lim_yn = cy / fy + 0.3 * tan_fovy
    tx = z * torch.minimum(lim_xp, torch.maximum(-lim_xn, x / z))
    ty = z * torch.minimum(lim_yp, torch.maximum(-lim_yn, y / z))
    zeros = torch.zeros_like(z)
    J = torch.stack([fx / z, zeros, -fx * tx / z ** 2, zeros, fy / z, -fy * ty / z ** 2], dim=-1).reshape(-1, 2, 3)
    cov2d = J @ covc @ J.transpose(1, 2)
    cov2d = cov2d + eps2d * torch.eye(2, dtype=means.dtype)
    det = cov2d[:, 0, 0] * cov2d[:, 1, 1] - cov2d[:, 0, 1] * cov2d[:, 1, 0]
    conic = torch.stack([cov2d[:, 1, 1] / det, -cov2d[:, 0, 1] / det, cov2d[:, 0, 0] / det], dim=-1)
    mean2d = torch.stack([fx * x / z + cx, fy * y / z + cy], dim=-1)
    return mean2d, z, conic


def sh_color(means, campos, sh):
    d = means - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = SH_C0 * sh[:, 0] + SH_C1 * (-y * sh[:, 1] + z * sh[:, 2] - x * sh[:, 3])
    return torch.clamp_min(c + 0.5, 0.0)


def render_dense(means, quats, scales, opacities, sh, viewmats, Ks, W, H, vis_mask, radii, tile_size=16):
    """vis_mask [C,N] bool and radii [C,N] int come from the (non-differentiable) culling.
    Returns rgb [C,H,W,3], alpha [C,H,W,1]."""
    Cn = viewmats.shape[0]
    dt = means.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt) + 0.5, torch.arange(W, dtype=dt) + 0.5, indexing="ij")
    px = xs.reshape(-1); py = ys.reshape(-1)
    ptx = torch.div(xs.reshape(-1) - 0.5, tile_size, rounding_mode="floor")
    pty = torch.div(ys.reshape(-1) - 0.5, tile_size, rounding_mode="floor")
    tw = math.ceil(W / tile_size); th = math.ceil(H / tile_size)
    out_rgb, out_a = [], []
    c2w = torch.inverse(viewmats)
    for c in range(Cn):
        idx = torch.nonzero(vis_mask[c]).reshape(-1)
        m2, depth, conic = project(means[idx], quats[idx], scales[idx], viewmats[c], Ks[c], W, H)
        col = sh_color(means[idx], c2w[c, :3, 3], sh[idx])
        op = opacities[idx]
        # depth order; ties keep gaussian-index order (stable), like the stable radix sort
        # on float32 depth bits: sort on the float32-rounded depth to reproduce ties
        d32 = depth.detach().to(torch.float32)
        order = torch.sort(d32, stable=True).indices
        m2, conic, col, op, rad = m2[order], conic[order], col[order], op[order], radii[c][idx][order].to(dt)
        # tile rectangle of every gaussian (non differentiable)
        m2d = m2.detach()
        x0 = torch.clamp(torch.floor((m2d[:, 0] - rad) / tile_size), 0, tw)
        x1 = torch.clamp(torch.ceil((m2d[:, 0] + rad) / tile_size), 0, tw)
        y0 = torch.clamp(torch.floor((m2d[:, 1] - rad) / tile_size), 0, th)
        y1 = torch.clamp(torch.ceil((m2d[:, 1] + rad) / tile_size), 0, th)
        in_rect = (ptx[:, None] >= x0) & (ptx[:, None] < x1) & (pty[:, None] >= y0) & (pty[:, None] < y1)
        dx = m2[None, :, 0] - px[:, None]; dy = m2[None, :, 1] - py[:, None]
        sigma = 0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) + conic[None, :, 1] * dx * dy
        alpha = torch.clamp_max(op[None] * torch.exp(-sigma), 0.999)
        valid = in_rect & (sigma >= 0) & (alpha >= 1.0 / 255.0)
        a = torch.where(valid, alpha, torch.zeros_like(alpha))
        nextT = torch.cumprod(1 - a, dim=1)
        stop = nextT <= 1e-4
        stop = torch.cummax(stop.to(torch.int8), dim=1).values.bool()
        a = torch.where(stop, torch.zeros_like(a), a)
        Tincl = torch.cumprod(1 - a, dim=1)
        Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], dim=1)
        w = a * Texcl
        rgb = w @ col
        Tfin = Tincl[:, -1] if Tincl.shape[1] else torch.ones_like(px)
        out_rgb.append(rgb.reshape(H, W, 3)); out_a.append((1 - Tfin).reshape(H, W, 1))
    return torch.stack(out_rgb), torch.stack(out_a)


def gaussian_window(dtype=torch.float64):
    g = torch.exp(-0.5 * ((torch.arange(11, dtype=torch.float64) - 5) / 1.5) ** 2)
    g = (g / g.sum()).to(torch.float32).to(dtype)  # torchmetrics builds the window in float32 [U]
    return g


def ssim_mean(x, y):
    """torchmetrics SSIM(data_range=1) restated [U]: x,y [H,W,3] -> scalar (mean over interior)."""
    g = gaussian_window(x.dtype)
    k2 = (g[:, None] * g[None, :])[None, None].repeat(3, 1, 1, 1)
    X = x.permute(2, 0, 1)[None]; Y = y.permute(2, 0, 1)[None]
    X = torch.nn.functional.pad(X, (5, 5, 5, 5), mode="reflect")
    Y = torch.nn.functional.pad(Y, (5, 5, 5, 5), mode="reflect")
    inp = torch.cat([X, Y, X * X, Y * Y, X * Y])
    o = torch.nn.functional.conv2d(inp, k2, groups=3)
    mx, my, exx, eyy, exy = o[0:1], o[1:2], o[2:3], o[3:4], o[4:5]
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    sxx = torch.clamp(exx - mx * mx, min=0); syy = torch.clamp(eyy - my * my, min=0); sxy = exy - mx * my   # torchmetrics
    s = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
    s = s[..., 5:-5, 5:-5]
    return s.reshape(1, -1).mean()


def view_loss(render, gt, opacities, scales, ssim_fac=0.2, opac_fac=0.01, scale_fac=0.01):
    """starster/gs.py:126-136 compute_loss for one view."""
    l1 = (gt - render).abs().mean()
    ss = 1 - ssim_mean(gt, render)
    loss = l1 * (1 - ssim_fac) + ss * ssim_fac
    loss = loss + opac_fac * torch.sigmoid(opacities).abs().mean()
    loss = loss + scale_fac * torch.exp(scales).abs().mean()
    return loss
