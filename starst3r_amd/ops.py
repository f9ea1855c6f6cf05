"""torch-tensor front-end of the C ABI (include/st3r.h).

PyTorch is plumbing here: it owns device memory and the stream; every function below hands
raw device pointers to libst3r_hip.so.  All tensors must live on the context's GPU.
"""
import ctypes as C
import math

import torch

from . import _lib

SPLAT = 12
TILE = 16


class Context:
    """One st3r_ctx per (process, GPU): owns the grow-only scratch arena."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.St3rError(f"the st3r hot path needs a GPU device, got {device!r} (no CPU fallback)")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._h = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib().st3r_ctx_create(idx, C.byref(self._h)))
        import os
        if os.environ.get("ST3R_DEBUG_FLAGS"):   # kernel A/B switches of st3r_ctx_set_debug (tools/, profiling runs)
            _lib.check(_lib.lib().st3r_ctx_set_debug(self._h, int(os.environ["ST3R_DEBUG_FLAGS"])))

    @property
    def handle(self):
        return self._h

    def arena_bytes(self):
        return int(_lib.lib().st3r_ctx_arena_bytes(self._h))

    def release_scratch(self):
        """Give the grow-only scratch arena back to the device allocator (st3r_ctx_release_scratch); the context stays
        valid and allocates again on its next call.  Raises what `settle` would raise."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().st3r_ctx_release_scratch(self._h))

    def close(self):
        if self._h:
            _lib.lib().st3r_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def get_context(device):
    d = torch.device(device)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    if idx not in _contexts:
        _contexts[idx] = Context(torch.device("cuda", idx))
    return _contexts[idx]


def release_scratch():
    """release_scratch() of every cached per-GPU context of this process (e.g. before handing the GPU to Mast3r
    inference or to another process)."""
    for ctx in _contexts.values():
        ctx.release_scratch()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"expected contiguous cuda {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()} "
                         f"device={t.device}")
    return C.c_void_p(t.data_ptr())


def _f(t):
    return t.detach().to(torch.float32).contiguous()


def tile_grid(W, H):
    return math.ceil(W / TILE), math.ceil(H / TILE)


def camera_positions(viewmats):
    """inverse(viewmats)[:, :3, 3] (gsplat: camtoworlds = torch.inverse(viewmats))."""
    return torch.inverse(viewmats)[:, :3, 3].contiguous()


def sh_stride_of(sh):
    return int(sh.numel() // sh.shape[0])


def project_sh(ctx, means, quats, scales, opacities, sh, viewmats, Ks, campos, W, H, reg_sums=None,
               eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0, out=None):
    """out: optional (splats [Cn*N,12], tiles int32 [Cn*N]) buffers to write into (steady-state loops)."""
    N, Cn = means.shape[0], viewmats.shape[0]
    if out is not None:
        splats, tiles = out
    else:
        splats = torch.empty((Cn * N, SPLAT), dtype=torch.float32, device=means.device)
        tiles = torch.empty((Cn * N,), dtype=torch.int32, device=means.device)
    _lib.check(_lib.lib().st3r_gs_project_sh(
        ctx.handle, _stream(), N, Cn, _p(means), _p(quats), _p(scales), _p(opacities), _p(sh), sh_stride_of(sh),
        _p(viewmats), _p(Ks), _p(campos), W, H, TILE, eps2d, near, far, radius_clip, _p(splats),
        _p(tiles, torch.int32), _p(reg_sums, torch.float64)))
    return splats, tiles


def isect_scan(ctx, tiles):
    """st3r_gs_isect_scan on its own: inclusive prefix sum of int32 counts -> (cum, total on the host)."""
    cum = torch.empty_like(tiles)
    n = C.c_int64(0)
    _lib.check(_lib.lib().st3r_gs_isect_scan(ctx.handle, _stream(), tiles.numel(), _p(tiles, torch.int32),
                                             _p(cum, torch.int32), C.byref(n)))
    return cum, n.value


def isect(ctx, splats, tiles, N, Cn, W, H):
    tw, th = tile_grid(W, H)
    cum = torch.empty_like(tiles)
    n = C.c_int64(0)
    _lib.check(_lib.lib().st3r_gs_isect_scan(ctx.handle, _stream(), N * Cn, _p(tiles, torch.int32),
                                             _p(cum, torch.int32), C.byref(n)))
    n = n.value
    ids = torch.empty((n,), dtype=torch.int64, device=splats.device)
    flat = torch.empty((n,), dtype=torch.int32, device=splats.device)
    _lib.check(_lib.lib().st3r_gs_isect_emit(ctx.handle, _stream(), N, Cn, _p(splats), _p(cum, torch.int32), TILE, tw,
                                             th, n, _p(ids, torch.int64), _p(flat, torch.int32)))
    return cum, ids, flat


def radix_sort_pairs(ctx, keys, vals, begin_bit, end_bit):
    """Stable ascending radix sort of (key, int32 value) pairs on key bits [begin_bit, end_bit).  keys: int32 or
    int64 tensor holding UNSIGNED 32/64-bit patterns; vals int32 or None.  Inputs are left untouched."""
    assert keys.dtype in (torch.int32, torch.int64)
    kb = 4 if keys.dtype == torch.int32 else 8
    ko = torch.empty_like(keys)
    vo = torch.empty_like(vals) if vals is not None else None
    _lib.check(_lib.lib().st3r_radix_sort_pairs(ctx.handle, _stream(), kb, keys.numel(), begin_bit, end_bit,
                                                _p(keys, keys.dtype), _p(vals, torch.int32), _p(ko, keys.dtype),
                                                _p(vo, torch.int32)))
    return ko, vo


def sort_pairs(ctx, ids, flat, end_bit):
    ids_in, flat_in = ids.clone(), flat.clone()
    ids_out, flat_out = torch.empty_like(ids), torch.empty_like(flat)
    _lib.check(_lib.lib().st3r_gs_sort(ctx.handle, _stream(), ids.numel(), end_bit, _p(ids_in, torch.int64),
                                       _p(flat_in, torch.int32), _p(ids_out, torch.int64),
                                       _p(flat_out, torch.int32)))
    return ids_out, flat_out


def offsets(ctx, ids_sorted, Cn, W, H):
    tw, th = tile_grid(W, H)
    off = torch.empty((Cn, th, tw), dtype=torch.int32, device=ids_sorted.device)
    _lib.check(_lib.lib().st3r_gs_offsets(ctx.handle, _stream(), ids_sorted.numel(), _p(ids_sorted, torch.int64), Cn,
                                          tw, th, _p(off, torch.int32)))
    return off


def blend_fwd(ctx, splats, off, flat, Cn, W, H):
    tw, th = tile_grid(W, H)
    dev = splats.device
    rgb = torch.empty((Cn, H, W, 3), dtype=torch.float32, device=dev)
    alpha = torch.empty((Cn, H, W, 1), dtype=torch.float32, device=dev)
    last = torch.empty((Cn, H, W), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().st3r_gs_blend_fwd(ctx.handle, _stream(), Cn, W, H, TILE, tw, th, _p(splats),
                                            _p(off, torch.int32), _p(flat, torch.int32), flat.numel(), _p(rgb),
                                            _p(alpha), _p(last, torch.int32)))
    return rgb, alpha, last


def blend_bwd(ctx, splats, off, flat, alpha, last, v_rgb, v_alpha, cum, Cn, W, H):
    tw, th = tile_grid(W, H)
    v_splats = torch.empty_like(splats)
    _lib.check(_lib.lib().st3r_gs_blend_bwd(ctx.handle, _stream(), Cn, W, H, TILE, tw, th, _p(splats),
                                            _p(off, torch.int32), _p(flat, torch.int32), flat.numel(), _p(alpha),
                                            _p(last, torch.int32), _p(v_rgb), _p(v_alpha), _p(cum, torch.int32),
                                            splats.shape[0], _p(v_splats)))
    return v_splats


def project_sh_bwd(ctx, means, quats, scales, opacities, sh, viewmats, Ks, campos, W, H, splats, v_splats,
                   reg_views=0.0, opac_fac=0.0, scale_fac=0.0, eps2d=0.3, out=None):
    N, Cn = means.shape[0], viewmats.shape[0]
    grads = out if out is not None else torch.empty((23 * N,), dtype=torch.float32, device=means.device)
    _lib.check(_lib.lib().st3r_gs_project_sh_bwd(
        ctx.handle, _stream(), N, Cn, _p(means), _p(quats), _p(scales), _p(opacities), _p(sh), sh_stride_of(sh),
        _p(viewmats), _p(Ks), _p(campos), W, H, eps2d, _p(splats), _p(v_splats), reg_views, opac_fac, scale_fac,
        _p(grads)))
    return grads


def split_grads(grads, N):
    """views into the block layout means[3N] quats[4N] scales[3N] opacities[N] sh4[12N]"""
    return dict(means=grads[0:3 * N].view(N, 3), quats=grads[3 * N:7 * N].view(N, 4),
                scales=grads[7 * N:10 * N].view(N, 3), opacities=grads[10 * N:11 * N],
                sh=grads[11 * N:23 * N].view(N, 4, 3))


def loss_l1_ssim(ctx, render, gt, w_l1, w_ssim, want_grad=True):
    Cn, H, W = render.shape[0], render.shape[1], render.shape[2]
    sums = torch.empty((Cn, 2), dtype=torch.float64, device=render.device)
    v = torch.empty_like(render) if want_grad else None
    _lib.check(_lib.lib().st3r_loss_l1_ssim(ctx.handle, _stream(), Cn, H, W, _p(render), _p(gt), w_l1, w_ssim,
                                            _p(sums, torch.float64), _p(v)))
    return sums, v


def gt_moments(ctx, gt):
    """(conv(gt), conv(gt^2)) of SSIM's 11 x 11 window for the images gt [C,H,W,3]: [C,H,W,3,2], computed once per
    training call (st3r_loss_gt_moments)."""
    Cn, H, W = gt.shape[0], gt.shape[1], gt.shape[2]
    mom = torch.empty((Cn, H, W, 3, 2), dtype=torch.float32, device=gt.device)
    _lib.check(_lib.lib().st3r_loss_gt_moments(ctx.handle, _stream(), Cn, H, W, _p(gt), _p(mom)))
    return mom


def set_gt_moments(ctx, gt, moments):
    """Register (gt, moments) with the ctx -- the loss kernels then read the moments instead of convolving gt -- or clear
    the registration (gt = None).  The caller keeps both tensors alive and unchanged while registered."""
    if gt is None or moments is None:
        _lib.check(_lib.lib().st3r_ctx_set_gt_moments(ctx.handle, None, None, 0, 0, 0))
        ctx._gtm_keep = None
        return
    assert gt.is_contiguous() and moments.is_contiguous() and moments.shape == (*gt.shape, 2)
    _lib.check(_lib.lib().st3r_ctx_set_gt_moments(ctx.handle, _p(gt), _p(moments), gt.shape[0], gt.shape[1], gt.shape[2]))
    ctx._gtm_keep = (gt, moments)


def adam_step(ctx, params, grads, m, v, lr, b1, b2, eps, step):
    """params: dict with means, quats, scales, opacities, shN (updated in place)."""
    N = params["means"].shape[0]
    sh = params["shN"]
    _lib.check(_lib.lib().st3r_adam_step(ctx.handle, _stream(), N, _p(params["means"]), _p(params["quats"]),
                                         _p(params["scales"]), _p(params["opacities"]), _p(sh), sh_stride_of(sh),
                                         _p(grads), _p(m), _p(v), lr, b1, b2, eps, step))


def train_fwd_bwd(ctx, params, viewmats, Ks, campos, gt, W, H, ssim_fac, opac_fac, scale_fac, grads, loss_out,
                  want_stats=True):
    """want_stats=False: see train_step."""
    N, Cn = params["means"].shape[0], viewmats.shape[0]
    sh = params["shN"]
    stats = (C.c_int64 * 4)() if want_stats else None
    _lib.check(_lib.lib().st3r_gs_train_fwd_bwd(
        ctx.handle, _stream(), N, Cn, _p(params["means"]), _p(params["quats"]), _p(params["scales"]),
        _p(params["opacities"]), _p(sh), sh_stride_of(sh), _p(viewmats), _p(Ks), _p(campos), _p(gt), W, H, ssim_fac,
        opac_fac, scale_fac, _p(grads), _p(loss_out), stats))
    if not want_stats:
        return None
    return dict(n_visible=int(stats[0]), n_isects=int(stats[1]), arena_bytes=int(stats[2]), n_isects_ref=int(stats[3]))


def adam_step_range(ctx, params, grads, m, v, lr, b1, b2, eps, step, i0, i1, stage=None):
    """Adam on the scalars [i0, i1) of the 23N-float buffers (the piece a rank owns after a reduce-scatter); `stage`
    (23N floats, buffer order) receives the new parameter values of the piece."""
    N = params["means"].shape[0]
    sh = params["shN"]
    _lib.check(_lib.lib().st3r_adam_step_range(
        ctx.handle, _stream(), N, _p(params["means"]), _p(params["quats"]), _p(params["scales"]), _p(params["opacities"]),
        _p(sh), sh_stride_of(sh), _p(grads), _p(m), _p(v), lr, b1, b2, eps, step, int(i0), int(i1),
        _p(stage) if stage is not None else None))


def params_from_stage(ctx, params, stage, i0, i1, limit):
    """parameters of the scalars outside [i0, i1) and below `limit` <- stage (buffer order)."""
    N = params["means"].shape[0]
    sh = params["shN"]
    _lib.check(_lib.lib().st3r_params_from_stage(
        ctx.handle, _stream(), N, _p(params["means"]), _p(params["quats"]), _p(params["scales"]), _p(params["opacities"]),
        _p(sh), sh_stride_of(sh), _p(stage), int(i0), int(i1), int(limit)))


def train_step(ctx, params, viewmats, Ks, campos, gt, W, H, ssim_fac, opac_fac, scale_fac, grads, m, v, lr, b1, b2, eps,
               step, loss_out, want_stats=True):
    """One whole iteration in one C call: fwd/bwd -> gradient all-reduce over the ctx's RCCL communicator (if
    one is attached, see dist.attach_native_comm) -> Adam.  loss_out gets this rank's part of the loss.
    want_stats=False: no statistics come back and the call does not synchronise with the device in steady state (the
    record count stays on the device; see st3r_gs_train_fwd_bwd in include/st3r.h)."""
    N, Cn = params["means"].shape[0], viewmats.shape[0]
    sh = params["shN"]
    stats = (C.c_int64 * 4)() if want_stats else None
    _lib.check(_lib.lib().st3r_gs_train_step(
        ctx.handle, _stream(), N, Cn, _p(params["means"]), _p(params["quats"]), _p(params["scales"]),
        _p(params["opacities"]), _p(sh), sh_stride_of(sh), _p(viewmats), _p(Ks), _p(campos), _p(gt), W, H, ssim_fac,
        opac_fac, scale_fac, _p(grads), _p(m), _p(v), lr, b1, b2, eps, step, _p(loss_out), stats))
    if not want_stats:
        return None
    return dict(n_visible=int(stats[0]), n_isects=int(stats[1]), arena_bytes=int(stats[2]), n_isects_ref=int(stats[3]))


def dense_unproject(ctx, view_start, pixels, idxs, offsets, core, cam_rows, base_focals):
    """All dense pixels of all views (concatenated) -> world points [n,3] and own-camera depth [n]."""
    n, Cn, G = pixels.shape[0], cam_rows.shape[0], core.shape[1]
    pts = torch.empty((n, 3), dtype=torch.float32, device=cam_rows.device)
    z = torch.empty((n,), dtype=torch.float32, device=cam_rows.device)
    _lib.check(_lib.lib().st3r_dense_unproject(ctx.handle, _stream(), Cn, G, n, _p(view_start, torch.int32), _p(pixels),
                                               _p(idxs, torch.int32), _p(offsets), _p(core), _p(cam_rows),
                                               _p(base_focals), _p(pts), _p(z)))
    return pts, z


def dense_clean(ctx, view_start, sizes_hw, cam_rows, pts, zcam, conf, tol=0.001, bad_conf=0.0):
    """dust3r clean_pointcloud on the concatenated views; returns the cleaned confidences (new tensor)."""
    out = conf.clone()
    mx = int((sizes_hw[:, 0].long() * sizes_hw[:, 1].long()).max())
    _lib.check(_lib.lib().st3r_dense_clean(ctx.handle, _stream(), cam_rows.shape[0], mx, _p(view_start, torch.int32),
                                           _p(sizes_hw, torch.int32), _p(cam_rows), _p(pts), _p(zcam), tol, bad_conf,
                                           _p(out)))
    return out


def canon_view(ctx, ptmaps, confs, subsample):
    """Mast3r canonical_view(mode='avg-angle') [U]: ptmaps [n,H,W,3], confs [n,H,W] -> canon [H,W,3], canon2 [H,W],
    cconf [H,W]."""
    n, H, W = confs.shape
    canon = torch.empty((H, W, 3), device=confs.device); canon2 = torch.empty((H, W), device=confs.device)
    cconf = torch.empty((H, W), device=confs.device)
    _lib.check(_lib.lib().st3r_canon_view(ctx.handle, _stream(), n, H, W, subsample, _p(ptmaps), _p(confs), _p(canon),
                                          _p(canon2), _p(cconf)))
    return canon, canon2, cconf


def focal_weiszfeld(ctx, canon, pp, min_focal=0.5, max_focal=3.5):
    """dust3r estimate_focal_knowing_depth(focal_mode='weiszfeld') [U] -> tensor [1] on the device."""
    H, W = canon.shape[:2]
    out = torch.empty(1, device=canon.device)
    _lib.check(_lib.lib().st3r_focal_weiszfeld(ctx.handle, _stream(), H, W, _p(canon), float(pp[0]), float(pp[1]),
                                               min_focal, max_focal, _p(out)))
    return out


def focal_weiszfeld_batch(ctx, canons, pp, min_focal=0.5, max_focal=3.5):
    """The same for a stack of images of one size in one launch: canons [n,H,W,3] -> tensor [n]."""
    n, H, W = canons.shape[:3]
    out = torch.empty(n, device=canons.device)
    _lib.check(_lib.lib().st3r_focal_weiszfeld_batch(ctx.handle, _stream(), n, H, W, _p(canons), float(pp[0]),
                                                     float(pp[1]), min_focal, max_focal, _p(out)))
    return out


def anchor_offsets(ctx, canon2, xy, subsample):
    """Mast3r anchor_depth_offsets [U] for one list of correspondence pixels xy [n,2] -> (idx int32 [n], off [n])."""
    H, W = canon2.shape
    n = xy.shape[0]
    idx = torch.empty(n, dtype=torch.int32, device=canon2.device); off = torch.empty(n, device=canon2.device)
    _lib.check(_lib.lib().st3r_anchor_offsets(ctx.handle, _stream(), n, H, W, subsample, _p(canon2), _p(xy),
                                              _p(idx, torch.int32), _p(off)))
    return idx, off


def raster_train(ctx, records, N, Cn, gt, W, H, ssim_fac, v_records, loss_out):
    """Middle phase of the Gaussian-sharded mode: records [Cn*N,12] of ALL Gaussians for this rank's views ->
    v_records (same shape), this rank's image loss."""
    stats = (C.c_int64 * 4)()
    _lib.check(_lib.lib().st3r_gs_raster_train(ctx.handle, _stream(), N, Cn, _p(records), _p(gt), W, H, ssim_fac,
                                               _p(v_records), _p(loss_out), stats))
    return dict(n_isects=int(stats[1]), arena_bytes=int(stats[2]))


def mcmc_relocate(ctx, params, m, v, min_opacity, seed, step, want_count=True):
    """In place on params (means, quats, scales, opacities, sh0 or None, shN) and on the fused Adam moments
    m, v ([23N] blocks, or None).  Returns the number of relocated Gaussians (None if not wanted: no sync)."""
    N = params["means"].shape[0]
    sh = params["shN"]; sh0 = params.get("sh0")
    n_dead = C.c_int64(0)
    _lib.check(_lib.lib().st3r_mcmc_relocate(
        ctx.handle, _stream(), N, _p(params["means"]), _p(params["quats"]), _p(params["scales"]),
        _p(params["opacities"]), None if sh0 is None else _p(sh0), _p(sh), sh_stride_of(sh),
        None if m is None else _p(m), None if v is None else _p(v), min_opacity, seed, step,
        C.byref(n_dead) if want_count else None))
    return int(n_dead.value) if want_count else None


def mcmc_add(ctx, params, N, n_new, min_opacity, seed, step):
    """params hold N + n_new rows; rows [N, N + n_new) are filled in place."""
    sh = params["shN"]; sh0 = params.get("sh0")
    assert params["means"].shape[0] >= N + n_new
    _lib.check(_lib.lib().st3r_mcmc_add(
        ctx.handle, _stream(), N, n_new, _p(params["means"]), _p(params["quats"]), _p(params["scales"]),
        _p(params["opacities"]), None if sh0 is None else _p(sh0), _p(sh), sh_stride_of(sh), min_opacity, seed, step))


def mcmc_noise(ctx, params, scaler, seed, step, row_offset=0):
    """row_offset: global row of params' first Gaussian (a shard of the set draws the noise of its own rows)."""
    N = params["means"].shape[0]
    _lib.check(_lib.lib().st3r_mcmc_noise_rows(ctx.handle, _stream(), N, int(row_offset), _p(params["means"]),
                                               _p(params["quats"]), _p(params["scales"]), _p(params["opacities"]),
                                               scaler, seed, step))


def peek(ctx, which, count, dtype=torch.int32):
    """Copy `count` elements of a ctx scratch buffer (see st3r_ctx_peek) into a new tensor (tests only)."""
    out = torch.empty((count,), dtype=dtype, device=ctx.device)
    _lib.check(_lib.lib().st3r_ctx_peek(ctx.handle, _stream(), which, _p(out, dtype), out.numel() * out.element_size()))
    return out


EXCHANGE_FORMS = ("allreduce", "ranges", "rs_ag", "direct")   # ST3R_EXCHANGE_* of include/st3r.h, in order


def set_exchange(ctx, form):
    """The form the gradient exchange inside train_step takes from now on (a setting of the ctx; all ranks must agree):
    'allreduce' | 'ranges' | 'rs_ag' | 'direct' (csrc/comm.hip).  Returns the previous form."""
    prev = get_exchange(ctx)
    _lib.check(_lib.lib().st3r_comm_set_exchange(ctx.handle, EXCHANGE_FORMS.index(form)))
    return prev


def get_exchange(ctx):
    f = C.c_int(-1)
    _lib.check(_lib.lib().st3r_comm_get_exchange(ctx.handle, C.byref(f)))
    return EXCHANGE_FORMS[f.value]


def allgather_pieces(ctx, buf):
    """In-place all-gather of the ranks' pieces of a [23N] buffer (the partition of the 'rs_ag' exchange): replicates
    Adam moments that were maintained piece-wise.  No-op without a communicator."""
    _lib.check(_lib.lib().st3r_comm_allgather_pieces(ctx.handle, _stream(), _p(buf), buf.numel()))


def settle(ctx):
    """Wait for the record count of the last asynchronous training step; raises St3rError (code -3) if that step outgrew
    its buffers (its Adam update was skipped on the device), code -5 if the last exchanged step failed on another rank
    (no rank applied it)."""
    _lib.check(_lib.lib().st3r_ctx_settle(ctx.handle))


def set_debug(ctx, flags):
    """bit 0: blend forward without the per-quadrant relevance test (tests only)."""
    _lib.check(_lib.lib().st3r_ctx_set_debug(ctx.handle, int(flags)))


STAGES = ("project", "scan", "emit", "sort", "offsets", "blend_fwd", "loss", "blend_bwd", "project_bwd", "adam",
          "sort_depth")


def set_profiling(ctx, enable, only=None):
    """HIP-event timing of the stages of the fused steps; `only` = a stage name: time just that stage."""
    code = 0 if not enable else (1 if only is None else 2 + STAGES.index(only))
    _lib.check(_lib.lib().st3r_ctx_set_profiling(ctx.handle, code))


def stage_ms(ctx):
    """-> {stage: (total_ms, samples)} accumulated since the last call (synchronises the device)."""
    n = 11  # ST3R_NUM_STAGES
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
    _lib.check(_lib.lib().st3r_ctx_get_stage_ms(ctx.handle, ms, cnt))
    return {_lib.lib().st3r_stage_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}


def render(ctx, params, viewmats, Ks, campos, W, H):
    N, Cn = params["means"].shape[0], viewmats.shape[0]
    sh = params["shN"]
    dev = params["means"].device
    rgb = torch.empty((Cn, H, W, 3), dtype=torch.float32, device=dev)
    alpha = torch.empty((Cn, H, W, 1), dtype=torch.float32, device=dev)
    stats = (C.c_int64 * 4)()
    _lib.check(_lib.lib().st3r_gs_render(
        ctx.handle, _stream(), N, Cn, _p(params["means"]), _p(params["quats"]), _p(params["scales"]),
        _p(params["opacities"]), _p(sh), sh_stride_of(sh), _p(viewmats), _p(Ks), _p(campos), W, H, _p(rgb), _p(alpha),
        stats))
    return rgb, alpha, dict(n_isects=int(stats[1]))


def rasterization(ctx, means, quats, scales, opacities, colors, viewmats, Ks, width, height, want_info=True):
    """Stage-by-stage rasterization with caller-owned outputs and the gsplat-style `info` dict
    (packed arrays, reference call: starster/gs.py:76-87).  Used by Scene.render_3dgs and tests."""
    N, Cn = means.shape[0], viewmats.shape[0]
    W, H = width, height
    campos = camera_positions(viewmats)
    splats, tiles = project_sh(ctx, means, quats, scales, opacities, colors, viewmats, Ks, campos, W, H)
    cum, ids, flat = isect(ctx, splats, tiles, N, Cn, W, H)
    tw, th = tile_grid(W, H)
    end_bit = 32 + (tw * th).bit_length() + Cn.bit_length()
    ids_s, flat_s = sort_pairs(ctx, ids, flat, end_bit)
    off = offsets(ctx, ids_s, Cn, W, H)
    rgb, alpha, last = blend_fwd(ctx, splats, off, flat_s, Cn, W, H)
    info = None
    if want_info:
        radii_dense = splats[:, 10].view(torch.int32)
        vis = radii_dense > 0
        pid = torch.nonzero(vis).reshape(-1)
        packed_of_dense = torch.cumsum(vis.to(torch.int64), 0) - 1
        sp = splats[pid]
        info = dict(
            camera_ids=torch.div(pid, N, rounding_mode="floor").to(torch.int32), gaussian_ids=(pid % N).to(torch.int32),
            radii=radii_dense[pid], means2d=sp[:, 0:2], depths=sp[:, 9], conics=sp[:, 3:6], opacities=sp[:, 2],
            colors=sp[:, 6:9], tile_width=tw, tile_height=th, tiles_per_gauss=tiles[pid], isect_ids=ids_s,
            flatten_ids=packed_of_dense[flat_s.long()].to(torch.int32), isect_offsets=off, width=W, height=H,
            tile_size=TILE, n_cameras=Cn,
            # dense-id extras used by the backward / tests
            _splats=splats, _flatten_ids_dense=flat_s, _last_ids=last, _isect_ids_unsorted=ids,
            _flatten_ids_dense_unsorted=flat, _packed_of_dense=packed_of_dense, _campos=campos, _cum_tiles=cum)
        global _LAST_INFO
        _LAST_INFO = info
    return rgb, alpha, info


_LAST_INFO = None


def last_info():
    """info dict of the most recent ops.rasterization(want_info=True) call (gsplat's `meta`)."""
    return _LAST_INFO
