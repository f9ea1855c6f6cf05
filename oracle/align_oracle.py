"""CPU restatement of the global-alignment optimiser (path B) -- TEST INFRASTRUCTURE ONLY.

Restates starster/reconstruct.py:116-457 `sparse_scene_optimizer_slam` with the reference's own
settings (reconstruct.py:61-69: lr1=0.07, niter1=500, lr2=0.014, niter2=200, opt_depth=False,
matching_conf_thr=5, shared_intrinsics=False) on the flat problem layout of
st3r_synth/synth_align.flatten().  Vectorised over views instead of python lists, torch CPU
autograd for the gradients (the HIP path uses hand-derived gradients; this file is what checks them).

PINNED: against golden vectors produced by the reference function itself
(tools/gen_align_goldens.py -> tests/golden/align_*.npz, tests/test_oracle_align.py).  The helper
formulas the reference star-imports from the absent mast3r package (gamma_loss, cosine_schedule,
make_pts3d, reproj2d, roma.unitquat_to_rotmat) are restated here and are pinned only
as far as those stubs go.
"""
import math

import numpy as np
import torch


def unitquat_to_rotmat(q):
    """roma convention (x, y, z, w); reconstruct.py:229"""
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def gamma_rho(d, gamma):
    """gamma_loss(gamma) applied to a distance d = ||x - y||  (App. A.5)"""
    if gamma == 1:
        return d
    off = (1 / gamma) ** (1 / (gamma - 1))
    return (d + off) ** gamma - off ** gamma


class Problem:
    def __init__(self, flat, dtype=torch.float32):
        t = lambda k, dt=dtype: torch.tensor(np.asarray(flat[k]), dtype=dt)
        self.C = int(flat["n_views"])
        self.imsizes = t("imsizes")                                # [C,2] (W,H)
        self.pps0 = t("pps") / self.imsizes                        # reconstruct.py:170
        self.base_focals = t("base_focals")
        cd = t("core_depth")
        # views of different image sizes: rows padded to the longest, core_len = the true lengths (the reference keeps
        # per-view lists, reconstruct.py:170-177); the median of a view is taken over its own values
        clen = [int(x) for x in np.asarray(flat["core_len"]).reshape(-1)] if "core_len" in flat else [cd.shape[1]] * cd.shape[0]
        self.core_len = clen
        self.median = torch.stack([cd[v, :clen[v]].median() for v in range(cd.shape[0])])   # :176
        self.core = cd / self.median[:, None]                      # :177
        self.anchor_pix = t("anchor_pix"); self.anchor_idx = t("anchor_idx", torch.int64)
        self.anchor_offset = t("anchor_offset"); self.anchor_img = t("anchor_img", torch.int64)
        self.root = int(flat["mst_root"]); self.edges = [tuple(int(v) for v in e) for e in np.asarray(flat["mst_edges"])]
        self.corr_a1 = t("corr_a1", torch.int64); self.corr_a2 = t("corr_a2", torch.int64); self.corr_conf = t("corr_conf")
        self.c2d_pix = t("c2d_pix"); self.c2d_a2 = t("c2d_a2", torch.int64); self.c2d_conf = t("c2d_conf")
        self.c2d_img1 = t("c2d_img1", torch.int64)
        self.dust_a1 = t("dust_a1", torch.int64); self.dust_tgt = t("dust_tgt"); self.dust_img2 = t("dust_img2", torch.int64)
        self.dust_conf = t("dust_conf")
        diags = self.imsizes.norm(dim=1)
        self.min_focals = 0.25 * diags; self.max_focals = 10 * diags  # :203-205
        self.dtype = dtype


def init_params(pb, prev=None):
    C, dt = pb.C, pb.dtype
    p = dict(pps=pb.pps0.clone(), log_focals=pb.base_focals.log().clone(),          # :200-201
             quats=torch.tensor([[0, 0, 0, 1.0]], dtype=dt).repeat(C, 1),           # :150-151 (xyzw)
             trans=torch.zeros(C, 3, dtype=dt), log_sizes=torch.zeros(C, dtype=dt))  # :152, :277
    if prev is not None:                                                            # warm start :408-415
        for k in p:
            n = min(len(prev[k]), C)
            p[k][:n] = torch.as_tensor(prev[k][:n], dtype=dt).reshape(p[k][:n].shape)
    return {k: v.requires_grad_(False) for k, v in p.items()}


def make_K_cam_depth(pb, p):
    """reconstruct.py:209-261"""
    C, dt = pb.C, pb.dtype
    focals = p["log_focals"].exp().clip(min=pb.min_focals, max=pb.max_focals)
    K = torch.eye(3, dtype=dt)[None].expand(C, 3, 3).clone()
    K[:, 0, 0] = K[:, 1, 1] = focals
    K[:, 0:2, 2] = p["pps"] * pb.imsizes
    sizes = p["log_sizes"].exp()
    global_scaling = 1 / sizes.min()
    z_cameras = sizes * pb.median * focals / pb.base_focals
    rel = torch.eye(4, dtype=dt)[None].expand(C, 4, 4).clone()
    rel[:, :3, :3] = unitquat_to_rotmat(torch.nn.functional.normalize(p["quats"], dim=1))
    rel[:, :3, 3] = p["trans"]
    tmp = [None] * C
    tmp[pb.root] = rel[pb.root]
    for i, j in pb.edges:
        tmp[j] = tmp[i] @ rel[j]
    tmp = torch.stack(tmp)
    ones = torch.ones((C, 1), dtype=dt)
    trans_offset = z_cameras.unsqueeze(1) * torch.cat((pb.imsizes / focals.unsqueeze(1) * (0.5 - p["pps"]), ones), dim=-1)
    new_trans = global_scaling * (tmp[:, :3, 3:4] - tmp[:, :3, :3] @ trans_offset.unsqueeze(-1))
    bottom = torch.tensor([0, 0, 0, 1.0], dtype=dt).view(1, 1, 4).expand(C, 1, 4)
    cam2w = torch.cat((torch.cat((tmp[:, :3, :3], new_trans), dim=2), bottom), dim=1)
    depth = global_scaling * (z_cameras[:, None] + (pb.core - 1) * (pb.median * sizes)[:, None])  # depth_mode 'add'
    return K, torch.linalg.inv(cam2w), cam2w, depth


def make_pts3d(pb, K, cam2w, depth):
    """App. A.5 make_pts3d over the concatenated anchors"""
    img = pb.anchor_img
    f = K[img, 0, 0]
    off = 1 + (pb.anchor_offset - 1) * (pb.base_focals[img] / f)
    z = depth[img, pb.anchor_idx] * off
    invK = torch.linalg.inv(K)
    hom = torch.cat((pb.anchor_pix, torch.ones_like(pb.anchor_pix[:, :1])), dim=-1)
    d = torch.diagonal(invK, dim1=1, dim2=2)[img]
    c = invK[img][:, :, 2] * torch.tensor([1.0, 1.0, 0.0], dtype=pb.dtype)
    pc = z.unsqueeze(-1) * (hom * d + c)
    return (cam2w[img, :3, :3] @ pc.unsqueeze(-1)).squeeze(-1) + cam2w[img, :3, 3]


def loss_3d(pb, pts, gamma=1.1):                        # reconstruct.py:325-353 (pix_loss = loss1, :118)
    if pb.corr_a1.numel() == 0:
        return torch.zeros((), dtype=pb.dtype)
    d = torch.linalg.norm(pts[pb.corr_a1] - pts[pb.corr_a2], dim=-1)
    return (pb.corr_conf @ gamma_rho(d, gamma)) / pb.corr_conf.sum()


def loss_2d(pb, K, w2cam, pts, gamma=0.4):              # reconstruct.py:355-369 (+ reproj2d, App. A.5; loss2, :119)
    if pb.c2d_a2.numel() == 0:
        return torch.zeros((), dtype=pb.dtype)
    P = K @ w2cam[:, :3]
    Pi = P[pb.c2d_img1]
    p = pts[pb.c2d_a2]
    r = (Pi[:, :, :3] @ p.unsqueeze(-1)).squeeze(-1) + Pi[:, :, 3]
    uv = (r[:, :2] / r[:, 2:3].clip(min=1e-3)).clip(min=-1000, max=2000)
    d = torch.linalg.norm(pb.c2d_pix - uv, dim=-1)
    return (pb.c2d_conf @ gamma_rho(d, gamma)) / pb.c2d_conf.sum()


def loss_dust3r(pb, cam2w, pts, gamma=1.1):             # reconstruct.py:311-323 (lossd, :120)
    if pb.dust_a1.numel() == 0:
        return torch.zeros((), dtype=pb.dtype)
    T = cam2w[pb.dust_img2]
    tgt = (T[:, :3, :3] @ pb.dust_tgt.unsqueeze(-1)).squeeze(-1) + T[:, :3, 3]
    d = torch.linalg.norm(pts[pb.dust_a1] - tgt, dim=-1)
    return (pb.dust_conf @ gamma_rho(d, gamma)) / pb.dust_conf.sum()


def cosine_schedule(alpha, lr_base, lr_end=0.0):
    return lr_end + (lr_base - lr_end) * (1 + np.cos(alpha * np.pi)) / 2


def linear_schedule(alpha, lr_base, lr_end=0.0):        # the other schedule the reference's helpers offer (App. A.5)
    return (1 - alpha) * lr_base + alpha * lr_end


def optimize_loop(pb, p, trainable, stage, lr_base, niter, loss_dust3r_w=0.01, losses=None, schedule=cosine_schedule,
                  gamma=None, gammad=1.1):
    """reconstruct.py:371-406"""
    gamma = (1.1 if stage == 1 else 0.4) if gamma is None else gamma
    for k, v in p.items():
        v.requires_grad_(k in trainable)
    core_trainable = "core_depth" in trainable                          # opt_depth (:437)
    pb.core.requires_grad_(core_trainable)
    opt = torch.optim.Adam([p[k] for k in ("pps", "log_focals", "quats", "trans", "log_sizes")] +
                           ([pb.core] if core_trainable else []), lr=1, weight_decay=0, betas=(0.9, 0.9))
    for it in range(niter or 1):
        K, w2cam, cam2w, depth = make_K_cam_depth(pb, p)
        pts = make_pts3d(pb, K, cam2w, depth)
        if niter == 0:
            break
        lr = schedule(it / niter, lr_base, 0)
        for g in opt.param_groups:
            g["lr"] = lr
        opt.zero_grad()
        main = loss_3d(pb, pts, gamma) if stage == 1 else loss_2d(pb, K, w2cam, pts, gamma)
        loss = main + loss_dust3r_w * loss_dust3r(pb, cam2w, pts, gammad)
        loss.backward()
        opt.step()
        with torch.no_grad():
            p["quats"] /= p["quats"].norm(dim=1, keepdim=True)       # :394-395
        lv = float(loss)
        if losses is not None:
            losses.append(lv)
        if lv != lv:
            break
    return dict(intrinsics=K.detach(), cam2w=cam2w.detach(), depthmaps=depth.detach(), pts3d=pts.detach())


def run(flat, lr1=0.07, niter1=500, lr2=0.014, niter2=200, prev=None, dtype=torch.float32, losses=None,
        schedule=cosine_schedule, gamma1=1.1, gamma2=0.4, gammad=1.1, opt_pp=True, opt_depth=False):
    """-> (result dict, params dict) like the reference's (res_fine or res_coarse, params_ret)."""
    pb = Problem(flat, dtype)
    if prev is not None and prev.get("core_depth") is not None:        # :414 -- the old views keep their core depth
        pc = torch.as_tensor(np.asarray(prev["core_depth"]), dtype=dtype)
        if pc.shape[1] == pb.core.shape[1]:
            n = min(pc.shape[0], pb.core.shape[0])
            pb.core = pb.core.clone(); pb.core[:n] = pc[:n]
    p = init_params(pb, prev)
    kw = dict(losses=losses, schedule=schedule, gammad=gammad)
    res = optimize_loop(pb, p, {"quats", "trans", "log_sizes"}, 1, lr1, niter1, gamma=gamma1, **kw)   # :418-427
    if niter2:
        train2 = {"quats", "trans", "log_sizes", "log_focals"} | ({"pps"} if opt_pp else set())     # :435-437
        if opt_depth:
            pb.core = pb.core.clone()
            train2.add("core_depth")
        res = optimize_loop(pb, p, train2, 2, lr2, niter2, gamma=gamma2, **kw)                      # :430-440
    params = {k: v.detach().numpy().copy() for k, v in p.items()}
    params["core_depth"] = pb.core.detach().numpy().copy()
    out = {k: v.numpy() for k, v in res.items()}
    out["core_len"] = np.asarray(pb.core_len, np.int64)      # true lengths of the padded core-depth rows
    return out, params
