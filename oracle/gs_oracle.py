"""ctypes front-end of oracle/gs_oracle.c (TEST INFRASTRUCTURE ONLY).

`rasterization()` mirrors the call the reference makes at starster/gs.py:76-87
(gsplat.rasterization(..., sh_degree=1), packed=True defaults) and returns the same
(render_colors, render_alphas, meta) triple as numpy arrays.  "Parity unpinned vs
upstream": gsplat is absent here, see gs_oracle.c.
"""
import ctypes as C
import math

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.lib_path("libgs_oracle.so"))
        _lib.gso_project_packed.restype = C.c_int64
        _lib.gso_isect_tiles.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def camera_positions(viewmats):
    """inverse(viewmats)[:, :3, 3]  (gsplat: camtoworlds = torch.inverse(viewmats) [U])"""
    return _f32(np.linalg.inv(np.asarray(viewmats, dtype=np.float64))[:, :3, 3])


def project_packed(means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    means, quats, scales, viewmats, Ks = map(_f32, (means, quats, scales, viewmats, Ks))
    N, Cn = means.shape[0], viewmats.shape[0]
    cap = N * Cn
    cam = np.empty(cap, np.int32); gid = np.empty(cap, np.int32); rad = np.empty(cap, np.int32)
    m2 = np.empty((cap, 2), np.float32); dep = np.empty(cap, np.float32); con = np.empty((cap, 3), np.float32)
    n = lib().gso_project_packed(C.c_int(N), C.c_int(Cn), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks),
                                 C.c_int(W), C.c_int(H), C.c_float(eps2d), C.c_float(near), C.c_float(far),
                                 C.c_float(radius_clip), _p(cam), _p(gid), _p(rad), _p(m2), _p(dep), _p(con))
    return dict(camera_ids=cam[:n].copy(), gaussian_ids=gid[:n].copy(), radii=rad[:n].copy(),
                means2d=m2[:n].copy(), depths=dep[:n].copy(), conics=con[:n].copy())


def sh_colors(camera_ids, gaussian_ids, means, campos, sh):
    means, campos, sh = map(_f32, (means, campos, sh))
    n = camera_ids.shape[0]
    stride = int(np.prod(sh.shape[1:]))
    out = np.empty((n, 3), np.float32)
    lib().gso_sh_colors(C.c_int64(n), _p(camera_ids), _p(gaussian_ids), _p(means), _p(campos), _p(sh),
                        C.c_int(stride), _p(out))
    return out


def isect_tiles(means2d, radii, depths, camera_ids, tile_size, tile_w, tile_h):
    n = radii.shape[0]
    tpg = np.empty(n, np.int32)
    ni = lib().gso_isect_tiles(C.c_int64(n), _p(means2d), _p(radii), _p(depths), _p(camera_ids), C.c_int(tile_size),
                               C.c_int(tile_w), C.c_int(tile_h), _p(tpg), None, None)
    ids = np.empty(ni, np.int64); flat = np.empty(ni, np.int32)
    ni2 = lib().gso_isect_tiles(C.c_int64(n), _p(means2d), _p(radii), _p(depths), _p(camera_ids),
                                C.c_int(tile_size), C.c_int(tile_w), C.c_int(tile_h), _p(tpg), _p(ids), _p(flat))
    assert ni == ni2
    return tpg, ids, flat


def sort_pairs(keys, vals):
    keys = np.ascontiguousarray(keys, np.int64).copy(); vals = np.ascontiguousarray(vals, np.int32).copy()
    lib().gso_sort_pairs(C.c_int64(keys.shape[0]), _p(keys), _p(vals))
    return keys, vals


def isect_offsets(sorted_ids, Cn, tile_w, tile_h):
    off = np.empty((Cn, tile_h, tile_w), np.int32)
    lib().gso_isect_offsets(C.c_int64(sorted_ids.shape[0]), _p(sorted_ids), C.c_int(Cn), C.c_int(tile_w),
                            C.c_int(tile_h), _p(off))
    return off


def blend_fwd(Cn, W, H, tile_size, means2d, conics, colors, opacities, offsets, flatten_ids, want_margin=False):
    tw, th = offsets.shape[2], offsets.shape[1]
    rgb = np.empty((Cn, H, W, 3), np.float32); alpha = np.empty((Cn, H, W, 1), np.float32)
    last = np.empty((Cn, H, W), np.int32)
    margin = np.empty((Cn, H, W), np.float32) if want_margin else None
    means2d, conics, colors, opacities = map(_f32, (means2d, conics, colors, opacities))
    lib().gso_blend_fwd(C.c_int(Cn), C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                        _p(means2d), _p(conics), _p(colors), _p(opacities), _p(offsets), _p(flatten_ids),
                        C.c_int64(flatten_ids.shape[0]), _p(rgb), _p(alpha), _p(last), _p(margin))
    return rgb, alpha, last, margin


def blend_bwd(Cn, W, H, tile_size, means2d, conics, colors, opacities, offsets, flatten_ids, alpha, last_ids,
              v_rgb, v_alpha=None):
    tw, th = offsets.shape[2], offsets.shape[1]
    n = means2d.shape[0]
    vm = np.zeros((n, 2), np.float64); vc = np.zeros((n, 3), np.float64)
    vcol = np.zeros((n, 3), np.float64); vo = np.zeros(n, np.float64)
    means2d, conics, colors, opacities, alpha, v_rgb = map(_f32, (means2d, conics, colors, opacities, alpha, v_rgb))
    if v_alpha is not None:
        v_alpha = _f32(v_alpha)
    lib().gso_blend_bwd(C.c_int(Cn), C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                        _p(means2d), _p(conics), _p(colors), _p(opacities), _p(offsets), _p(flatten_ids),
                        C.c_int64(flatten_ids.shape[0]), _p(alpha), _p(last_ids), _p(v_rgb), _p(v_alpha), _p(vm),
                        _p(vc), _p(vcol), _p(vo))
    return vm, vc, vcol, vo


def project_sh_bwd(N, camera_ids, gaussian_ids, means, quats, scales, sh, viewmats, Ks, campos, W, H, v_means2d,
                   v_conics, v_colors, v_opac_packed, eps2d=0.3):
    means, quats, scales, sh, viewmats, Ks, campos = map(_f32, (means, quats, scales, sh, viewmats, Ks, campos))
    stride = int(np.prod(sh.shape[1:]))
    v_means = np.zeros((N, 3)); v_quats = np.zeros((N, 4)); v_scales = np.zeros((N, 3))
    v_opac = np.zeros(N); v_sh = np.zeros((N, 4, 3))
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    v_means2d, v_conics, v_colors, v_opac_packed = map(f64, (v_means2d, v_conics, v_colors, v_opac_packed))
    lib().gso_project_sh_bwd(C.c_int64(camera_ids.shape[0]), C.c_int(N), _p(camera_ids), _p(gaussian_ids),
                             _p(means), _p(quats), _p(scales), _p(sh), C.c_int(stride), _p(viewmats), _p(Ks),
                             _p(campos), C.c_int(W), C.c_int(H), C.c_float(eps2d), _p(v_means2d), _p(v_conics),
                             _p(v_colors), _p(v_opac_packed), _p(v_means), _p(v_quats), _p(v_scales), _p(v_opac),
                             _p(v_sh))
    return dict(means=v_means, quats=v_quats, scales=v_scales, opacities=v_opac, sh=v_sh)


def l1_ssim(render, gt, w_l1=0.8, w_ssim=0.2, want_grad=True):
    render, gt = _f32(render), _f32(gt)
    H, W = render.shape[:2]
    l1 = C.c_double(); ss = C.c_double()
    vr = np.empty_like(render) if want_grad else None
    lib().gso_l1_ssim(C.c_int(H), C.c_int(W), _p(render), _p(gt), C.c_double(w_l1), C.c_double(w_ssim),
                      C.byref(l1), C.byref(ss), _p(vr))
    return l1.value, ss.value, vr


def adam(p, g, m, v, lr, b1, b2, eps, step):
    """In place on float32 arrays p, m, v."""
    assert p.dtype == np.float32 and m.dtype == np.float32 and v.dtype == np.float32
    g = _f32(g)
    lib().gso_adam(C.c_int64(p.size), _p(p), _p(g), _p(m), _p(v), C.c_double(lr), C.c_double(b1), C.c_double(b2),
                   C.c_double(eps), C.c_int(step))


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree=1,
                  tile_size=16, want_margin=False):
    """gsplat.rasterization restated (packed=True, classic, RGB, no background)."""
    assert sh_degree == 1
    viewmats = _f32(viewmats); Ks = _f32(Ks)
    Cn = viewmats.shape[0]
    pr = project_packed(means, quats, scales, viewmats, Ks, width, height)
    campos = camera_positions(viewmats)
    cols = sh_colors(pr["camera_ids"], pr["gaussian_ids"], means, campos, colors)
    opac = _f32(opacities)[pr["gaussian_ids"]]
    tw = math.ceil(width / float(tile_size)); th = math.ceil(height / float(tile_size))
    tpg, ids, flat = isect_tiles(pr["means2d"], pr["radii"], pr["depths"], pr["camera_ids"], tile_size, tw, th)
    ids_s, flat_s = sort_pairs(ids, flat)
    off = isect_offsets(ids_s, Cn, tw, th)
    rgb, alpha, last, margin = blend_fwd(Cn, width, height, tile_size, pr["means2d"], pr["conics"], cols, opac, off,
                                         flat_s, want_margin)
    meta = dict(pr)
    meta.update(opacities=opac, colors=cols, tile_width=tw, tile_height=th, tiles_per_gauss=tpg, isect_ids=ids_s,
                flatten_ids=flat_s, isect_offsets=off, width=width, height=height, tile_size=tile_size,
                n_cameras=Cn, last_ids=last, margin=margin, isect_ids_unsorted=ids, flatten_ids_unsorted=flat,
                campos=campos)
    return rgb, alpha, meta


def rasterization_backward(means, quats, scales, opacities, colors, viewmats, Ks, width, height, meta, alpha,
                           v_rgb, v_alpha=None):
    """Analytic backward of `rasterization` -> dict of per-parameter gradients (float64)."""
    N = np.asarray(means).shape[0]
    Cn = np.asarray(viewmats).shape[0]
    vm, vc, vcol, vo = blend_bwd(Cn, width, height, meta["tile_size"], meta["means2d"], meta["conics"],
                                 meta["colors"], meta["opacities"], meta["isect_offsets"], meta["flatten_ids"],
                                 alpha, meta["last_ids"], v_rgb, v_alpha)
    g = project_sh_bwd(N, meta["camera_ids"], meta["gaussian_ids"], means, quats, scales, colors, viewmats, Ks,
                       meta["campos"], width, height, vm, vc, vcol, vo)
    g["packed"] = dict(v_means2d=vm, v_conics=vc, v_colors=vcol, v_opacities=vo)
    return g
