#!/usr/bin/env python
"""Generates tests/golden/align_*.npz by running the REFERENCE's own optimiser
(/root/reference/starster/reconstruct.py:116-457 `sparse_scene_optimizer_slam`) in this container.

The reference file is executed from where it lies (never copied); the third-party names it
star-imports from the absent `mast3r.cloud_opt.sparse_ga` are stubbed with the helper formulas of
SURVEY.md App. A.5 (those helpers are therefore NOT pinned by these vectors -- only the
reference-owned parametrisation, losses, schedule, Adam(0.9,0.9) and quaternion renormalisation are).
Only numeric arrays are written out.  Run:  python tools/gen_align_goldens.py
"""
import copy
import importlib.util
import io
import contextlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/starster/reconstruct.py"


# ---------------- stubs for the absent third-party helpers ----------------
def unitquat_to_rotmat(q):  # roma convention: (x, y, z, w)
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def l1_loss(x, y):
    return torch.linalg.norm(x - y, dim=-1)


def gamma_loss(gamma, mul=1, offset=None, clip=np.inf):
    if offset is None:
        if gamma == 1:
            return l1_loss
        offset = (1 / gamma) ** (1 / (gamma - 1))

    def loss_func(x, y):
        return (mul * l1_loss(x, y).clip(max=clip) + offset) ** gamma - offset ** gamma
    return loss_func


def cosine_schedule(alpha, lr_base, lr_end=0):
    return lr_end + (lr_base - lr_end) * (1 + np.cos(alpha * np.pi)) / 2


def linear_schedule(alpha, lr_base, lr_end=0):
    return (1 - alpha) * lr_base + alpha * lr_end


def adjust_learning_rate_by_lr(optimizer, lr):
    for g in optimizer.param_groups:
        g["lr"] = lr * g["lr_scale"] if "lr_scale" in g else lr


def geotrf(T, p):
    return p @ T[:3, :3].T + T[:3, 3]


def reproj2d(P, p):
    r = p @ P[:, :3].T + P[:, 3]
    uv = r[:, :2] / r[:, 2:3].clip(min=1e-3)
    return uv.clip(min=-1000, max=2000)


def make_pts3d(anchors, K, cam2w, depthmaps, base_focals=None):
    focals = K[:, 0, 0]
    invK = torch.linalg.inv(K)
    out = []
    for img, (pixels, idxs, offsets) in anchors.items():
        if base_focals is not None:
            offsets = 1 + (offsets - 1) * (base_focals[img] / focals[img])
        z = depthmaps[img][idxs] * offsets
        hom = torch.cat((pixels, torch.ones_like(pixels[..., :1])), dim=-1)
        p = z.unsqueeze(-1) * (hom * invK[img].diag() + invK[img][:, 2] * torch.tensor([1.0, 1.0, 0.0]))
        out.append(geotrf(cam2w[img], p))
    return out


def load_reference():
    roma = types.ModuleType("roma"); roma.unitquat_to_rotmat = unitquat_to_rotmat
    sga = types.ModuleType("mast3r.cloud_opt.sparse_ga")
    from tqdm import tqdm
    for k, v in dict(copy=copy, torch=torch, nn=nn, F=F, np=np, tqdm=tqdm, roma=roma, inv=torch.linalg.inv,
                     geotrf=geotrf, to_numpy=lambda x: x.detach().cpu().numpy(), cosine_schedule=cosine_schedule,
                     adjust_learning_rate_by_lr=adjust_learning_rate_by_lr, gamma_loss=gamma_loss,
                     reproj2d=reproj2d, make_pts3d=make_pts3d).items():
        setattr(sga, k, v)
    mods = {"mast3r": types.ModuleType("mast3r"), "mast3r.cloud_opt": types.ModuleType("mast3r.cloud_opt"),
            "mast3r.cloud_opt.sparse_ga": sga, "dust3r": types.ModuleType("dust3r"),
            "dust3r.image_pairs": types.ModuleType("dust3r.image_pairs"), "starster": types.ModuleType("starster"),
            "starster.image": types.ModuleType("starster.image"), "roma": roma}
    mods["dust3r.image_pairs"].make_pairs = lambda *a, **k: None
    mods["starster.image"].prepare_images_for_mast3r = lambda *a, **k: None
    mods["starster"].__path__ = []
    sys.modules.update(mods)
    spec = importlib.util.spec_from_file_location("starster.reconstruct", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


from st3r_synth.synth_align import Slice, to_reference_inputs  # noqa: E402  (shared with the product's B1 tests)


def _to_f64(x):
    """float32 tensors of the reference-input structure -> float64 (same values), recursively."""
    if torch.is_tensor(x):
        return x.double() if x.dtype == torch.float32 else x
    if isinstance(x, Slice):
        return Slice(x.img1, x.slice1, x.img2, x.slice2, _to_f64(x.confs))
    if isinstance(x, dict):
        return {k: _to_f64(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_f64(v) for v in x)
    return x


def run_reference(ref, P, niter1, niter2, f64=False, **extra):
    """f64: the SAME reference function evaluated in float64 (its own `dtype` argument + torch's default dtype): the
    yardstick that tells how far a float32 trajectory -- the reference's included -- drifts from the exact one."""
    a = to_reference_inputs(P)
    dtype = torch.float32
    if f64:
        a = _to_f64(a); dtype = torch.float64
        torch.set_default_dtype(torch.float64)
    torch.manual_seed(0)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            imgs, coarse, fine, params = ref.sparse_scene_optimizer_slam(
                a["imgs"], a["subsample"], a["imsizes"], a["pps"], a["base_focals"], a["core_depth"], a["anchors"],
                a["corres"], a["corres2d"], a["preds_21"], a["canonical_paths"], a["mst"], cache_path=None,
                lr1=0.07, niter1=niter1, lr2=0.014, niter2=niter2, device="cpu", dtype=dtype,
                matching_conf_thr=5, shared_intrinsics=False,  # the reference's own settings, reconstruct.py:61-69
                **{"opt_depth": False, **extra})
    finally:
        torch.set_default_dtype(torch.float32)
    res = fine or coarse
    out = {}
    for k in ("pps", "log_focals", "quats", "trans", "log_sizes"):
        out["p_" + k] = np.stack([p.detach().numpy().reshape(-1) for p in params[k]])
    from st3r_synth.synth_align import pad_core_depth
    out["p_core_depth"] = pad_core_depth([p.detach().numpy() for p in params["core_depth"]])[0]   # padded if ragged
    out["intrinsics"] = res["intrinsics"].numpy(); out["cam2w"] = res["cam2w"].numpy()
    out["depthmaps"] = pad_core_depth([d.numpy() for d in res["depthmaps"]])[0]
    out["pts3d"] = np.concatenate([p.numpy() for p in res["pts3d"]])
    return out


def main():
    from st3r_synth import synth_align
    ref = load_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    configs = [("align_c2", dict(n_views=2, n_corr=400, seed=1)),
               ("align_c4_badpair", dict(n_views=4, n_corr=250, seed=2, bad_pair=True)),
               # one landscape, one portrait, one smaller landscape photo: core-depth vectors of 3072 / 3072 / 1728 values
               ("align_c3_mixed_sizes", dict(n_views=3, n_corr=300, seed=5, sizes=[(512, 384), (384, 512), (384, 288)])),
               # eight views (the view count of configs[1] and of bench.py's alignment line): a deeper spanning tree, 28 pairs;
               # small images keep the fixture small
               ("align_c8_small", dict(n_views=8, n_corr=120, seed=11, sizes=[(256, 192)] * 8))]
    only = sys.argv[1:]
    if not only or "align_c3_opts" in only:
        # non-default optimiser options of the same function (reconstruct.py:118-122): other robust losses, the linear
        # schedule, principal points frozen
        P = synth_align.make_problem(n_views=3, n_corr=300, seed=7, bad_pair=True)
        flat = synth_align.flatten(P)
        runs = {}
        opts = dict(loss1=gamma_loss(1.5), loss2=gamma_loss(0.6), lossd=gamma_loss(1), schedule=linear_schedule, opt_pp=False)
        for (n1, n2) in ((10, 0), (500, 0), (500, 200)):
            r = run_reference(ref, P, n1, n2, **opts)
            for k, v in r.items():
                runs[f"r{n1}_{n2}__{k}"] = v.astype(np.float32) if v.dtype.kind == "f" else v
            print("align_c3_opts", (n1, n2), "focals", r["intrinsics"][:, 0, 0])
        r = run_reference(ref, P, 500, 200, f64=True, **opts)
        for k, v in r.items():
            runs[f"f64_r500_200__{k}"] = v
        np.savez_compressed(os.path.join(out_dir, "align_c3_opts.npz"), **{"in__" + k: v for k, v in flat.items()}, **runs)
        print("wrote align_c3_opts", os.path.getsize(os.path.join(out_dir, "align_c3_opts.npz")) // 1024, "KiB")
    if not only or "align_c3_optdepth" in only:
        # opt_depth=True (the default of the function's signature, reconstruct.py:121; the reference's caller passes False):
        # the core depths are parameters of the second stage
        P = synth_align.make_problem(n_views=3, n_corr=300, seed=8, bad_pair=True)
        flat = synth_align.flatten(P)
        runs = {}
        for (n1, n2) in ((500, 1), (500, 10), (500, 200)):
            r = run_reference(ref, P, n1, n2, opt_depth=True)
            for k, v in r.items():
                runs[f"r{n1}_{n2}__{k}"] = v.astype(np.float32) if v.dtype.kind == "f" else v
            print("align_c3_optdepth", (n1, n2), "focals", r["intrinsics"][:, 0, 0])
        r = run_reference(ref, P, 500, 200, f64=True, opt_depth=True)
        for k, v in r.items():
            runs[f"f64_r500_200__{k}"] = v
        np.savez_compressed(os.path.join(out_dir, "align_c3_optdepth.npz"), **{"in__" + k: v for k, v in flat.items()}, **runs)
        print("wrote align_c3_optdepth", os.path.getsize(os.path.join(out_dir, "align_c3_optdepth.npz")) // 1024, "KiB")
    configs = [c for c in configs if not only or c[0] in only]
    for name, kw in configs:
        P = synth_align.make_problem(**kw)
        flat = synth_align.flatten(P)
        runs = {}
        for (n1, n2) in ((1, 0), (10, 0), (500, 0), (500, 1), (500, 200)):
            r = run_reference(ref, P, n1, n2)
            for k, v in r.items():
                runs[f"r{n1}_{n2}__{k}"] = v.astype(np.float32) if v.dtype.kind == "f" else v
            print(name, (n1, n2), "focals", r["intrinsics"][:, 0, 0])
        # float64 evaluation of the reference at the points where float32 trajectories are compared loosely
        for (n1, n2) in ((10, 0), (500, 0), (500, 200)):
            r = run_reference(ref, P, n1, n2, f64=True)
            for k, v in r.items():
                runs[f"f64_r{n1}_{n2}__{k}"] = v      # kept in float64
            print(name, (n1, n2), "float64 focals", r["intrinsics"][:, 0, 0])
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **{"in__" + k: v for k, v in flat.items()}, **runs)
        print("wrote", name, os.path.getsize(os.path.join(out_dir, name + ".npz")) // 1024, "KiB")
    # interp_se3 goldens (starster/utils.py is the only reference file importable as is)
    spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/starster/utils.py")
    u = importlib.util.module_from_spec(spec); spec.loader.exec_module(u)
    g = torch.Generator().manual_seed(3)
    def rand_se3():
        q = torch.randn(4, generator=g); q = q / q.norm()
        M = torch.eye(4); M[:3, :3] = unitquat_to_rotmat(q); M[:3, 3] = torch.randn(3, generator=g)
        return M
    A, B = rand_se3(), rand_se3()
    np.savez_compressed(os.path.join(out_dir, "interp_se3.npz"), A=A.numpy(), B=B.numpy(),
                        f025=u.interp_se3(A, B, 0.25).numpy(), f07=u.interp_se3(A, B, 0.7).numpy(),
                        path5=u.interp_se3_path(A, B, 5).numpy())
    print("wrote interp_se3")


if __name__ == "__main__":
    main()
