"""Host side of the global alignment (path B): prepares the flat problem on the device and runs
st3r_align_run.  Mirrors the reference's `sparse_scene_optimizer_slam`
(starster/reconstruct.py:116-457): same parameters (`optim_params` keys pps, log_focals, quats,
trans, log_sizes, core_depth), same defaults, same return pieces (intrinsics, cam2w, depthmaps, pts3d).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops

PARAM_KEYS = ("pps", "log_focals", "quats", "trans", "log_sizes")


def run(flat, lr1=0.07, niter1=500, lr2=0.014, niter2=200, prev_params=None, loss_dust3r_w=0.01, device="cuda:0",
        schedule=None, gamma1=1.1, gamma2=0.4, gammad=1.1, opt_pp=True, opt_depth=False):
    """flat: dict of numpy arrays (st3r_synth.synth_align.flatten layout).
    Returns (result, params): result has intrinsics [C,3,3], cam2w [C,4,4], depthmaps [C,G], pts3d [A,3],
    losses [niter1+niter2] (st3r_align_run stops updating after a NaN loss, like the reference's `break` at
    starster/reconstruct.py:397-398: the remaining entries stay 0); params holds the optimised parameters (and the normalised core_depth) so that a
    later call can warm start from them (reconstruct.py:408-415).
    schedule: callable (alpha, lr_base, lr_end) -> lr like the reference's `schedule` argument (reconstruct.py:122,
    385), evaluated here once per iteration (None: cosine_schedule inside the library); gamma1 / gamma2 / gammad: the
    exponents of loss1 / loss2 / lossd = gamma_loss(g) (:118-120); opt_pp (:121, 436); opt_depth (:121, 437): the
    (normalised) core depths are parameters of the second stage too -- `params["core_depth"]` then holds the optimised
    values and the depthmaps of the result use the values of the last iteration's start, like the reference's."""
    ctx = ops.get_context(device)
    dev = ctx.device
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev).contiguous()
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a).astype(np.int32), dtype=torch.int32, device=dev).contiguous()
    Cn = int(flat["n_views"])
    imsizes = f32(flat["imsizes"])
    base_focals = f32(flat["base_focals"])
    core_raw = f32(flat["core_depth"])
    G = core_raw.shape[1]
    # views of different image sizes have core-depth vectors of different lengths (the reference keeps per-view lists,
    # reconstruct.py:170-177, 276): the rows are padded to the longest, core_len holds the true lengths, the median of a
    # view is taken over its own values
    core_len = [int(x) for x in np.asarray(flat["core_len"]).reshape(-1)] if "core_len" in flat else [G] * Cn
    if all(n == G for n in core_len):
        median = core_raw.median(dim=1).values.contiguous()             # reconstruct.py:176
    else:
        median = torch.stack([core_raw[v, :core_len[v]].median() for v in range(Cn)]).contiguous()
    core = (core_raw / median[:, None]).contiguous()                    # :177
    diags = imsizes.norm(dim=1)
    min_f = (0.25 * diags).contiguous(); max_f = (10 * diags).contiguous()  # :203-205

    def weights(conf):
        c = f32(conf)
        return (c / c.sum()).contiguous() if c.numel() else c

    anchor_pix, anchor_idx = f32(flat["anchor_pix"]), i32(flat["anchor_idx"])
    anchor_off, anchor_img = f32(flat["anchor_offset"]), i32(flat["anchor_img"])
    corr_a1, corr_a2, corr_w = i32(flat["corr_a1"]), i32(flat["corr_a2"]), weights(flat["corr_conf"])
    c2d_pix, c2d_a2, c2d_img1 = f32(flat["c2d_pix"]), i32(flat["c2d_a2"]), i32(flat["c2d_img1"])
    c2d_w = weights(flat["c2d_conf"])
    dust_a1, dust_tgt, dust_img2 = i32(flat["dust_a1"]), f32(flat["dust_tgt"]), i32(flat["dust_img2"])
    dust_w = weights(flat["dust_conf"])
    edges = i32(np.asarray(flat["mst_edges"]).reshape(-1, 2))

    # parameters (reconstruct.py:150-152, 170, 200-201, 277); warm start splices the first n views (:408-415)
    P = dict(pps=(f32(flat["pps"]) / imsizes).contiguous(), log_focals=base_focals.log().contiguous(),
             quats=torch.tensor([[0, 0, 0, 1.0]], device=dev).repeat(Cn, 1).contiguous(),
             trans=torch.zeros(Cn, 3, device=dev), log_sizes=torch.zeros(Cn, device=dev))
    if prev_params is not None:
        for k in PARAM_KEYS:
            prev = prev_params[k]
            if isinstance(prev, (list, tuple)):   # the reference's format: a list of per-view nn.Parameters
                prev = torch.stack([torch.as_tensor(x).detach().reshape(-1) for x in prev])
            prev = prev.detach() if torch.is_tensor(prev) else torch.as_tensor(np.asarray(prev))
            prev = prev.to(dev, torch.float32)
            n = min(prev.shape[0], Cn)
            P[k][:n] = prev[:n].reshape(P[k][:n].shape)
        if prev_params.get("core_depth") is not None:   # reconstruct.py:414: the old views keep their core depth
            prev = prev_params["core_depth"]
            prev_len = prev_params.get("core_len")
            rows = list(prev) if isinstance(prev, (list, tuple)) else [prev[v] for v in range(prev.shape[0])]
            for v in range(min(len(rows), Cn)):                         # spliced view by view: lengths may differ
                r = torch.as_tensor(rows[v]).detach().reshape(-1).to(dev, torch.float32)
                if prev_len is not None:
                    r = r[:int(prev_len[v])]
                if r.numel() != core_len[v]:
                    raise ValueError(f"warm start: view {v} had {r.numel()} core depths, this call has {core_len[v]} "
                                     "(image size or subsampling changed between add_images calls)")
                core[v, :core_len[v]] = r
    work = torch.zeros(66 * Cn + 8, device=dev)
    cam = torch.empty(Cn, 24, device=dev)
    A = anchor_idx.numel()
    pts = torch.empty(A, 3, device=dev)
    losses = torch.zeros(max(niter1 + niter2, 1), device=dev)
    p = ops._p
    lr_host = None
    if schedule is not None:   # reconstruct.py:384-385: alpha = iter / niter, lr = schedule(alpha, lr_base, lr_end = 0)
        lrs = [float(schedule(it / n, lr, 0)) for lr, n in ((lr1, niter1), (lr2, niter2)) for it in range(n)]
        lr_host = np.ascontiguousarray(np.asarray(lrs + [0.0], np.float32))
    csr_off = csr_rows = depth_work = None
    if opt_depth and niter2 > 0:
        # rows of stage 2 (loss_2d rows, then regression rows) grouped by the core depth their anchor reads; the kernel
        # adds a core depth's rows in this (stable) order
        a_img = np.asarray(flat["anchor_img"]).astype(np.int64); a_idx = np.asarray(flat["anchor_idx"]).astype(np.int64)
        a_row = np.concatenate([np.asarray(flat["c2d_a2"]).astype(np.int64).reshape(-1),
                                np.asarray(flat["dust_a1"]).astype(np.int64).reshape(-1)])
        elem = a_img[a_row] * G + a_idx[a_row]
        order = np.argsort(elem, kind="stable")
        csr_rows = i32(order if order.size else np.zeros(1, np.int64))
        csr_off = i32(np.searchsorted(elem[order], np.arange(Cn * G + 1), side="left"))
        depth_work = torch.zeros(a_row.size + 3 * Cn * G, device=dev)
    _lib.check(_lib.lib().st3r_align_run_opts(
        ctx.handle, ops._stream(), Cn, G, A, p(imsizes), p(base_focals), p(median), p(core), p(min_f), p(max_f),
        p(anchor_pix), p(anchor_idx, torch.int32), p(anchor_off), p(anchor_img, torch.int32),
        corr_a1.numel(), p(corr_a1, torch.int32), p(corr_a2, torch.int32), p(corr_w),
        c2d_a2.numel(), p(c2d_pix), p(c2d_a2, torch.int32), p(c2d_img1, torch.int32), p(c2d_w),
        dust_a1.numel(), p(dust_a1, torch.int32), p(dust_tgt), p(dust_img2, torch.int32), p(dust_w),
        int(flat["mst_root"]), edges.shape[0], p(edges, torch.int32), lr1, niter1, lr2, niter2, loss_dust3r_w,
        p(P["pps"]), p(P["log_focals"]), p(P["quats"]), p(P["trans"]), p(P["log_sizes"]),
        p(work), work.numel(), p(cam), p(pts), p(losses),
        lr_host.ctypes.data if lr_host is not None else None, float(gamma1), float(gamma2), float(gammad), int(bool(opt_pp)),
        p(csr_off, torch.int32) if csr_off is not None else None, p(csr_rows, torch.int32) if csr_rows is not None else None,
        p(depth_work) if depth_work is not None else None, depth_work.numel() if depth_work is not None else 0))
    K = torch.zeros(Cn, 3, 3, device=dev)
    K[:, 0, 0] = K[:, 1, 1] = cam[:, 12]; K[:, 0, 2] = cam[:, 13]; K[:, 1, 2] = cam[:, 14]; K[:, 2, 2] = 1
    cam2w = torch.zeros(Cn, 4, 4, device=dev)
    cam2w[:, :3, :3] = cam[:, :9].reshape(Cn, 3, 3); cam2w[:, :3, 3] = cam[:, 9:12]; cam2w[:, 3, 3] = 1
    # (opt_depth: the results belong to the core depths as they were at the start of the last iteration)
    core_res = depth_work[-Cn * G:].reshape(Cn, G) if depth_work is not None else core
    depth = cam[:, 15:16] + cam[:, 16:17] * core_res
    res = dict(intrinsics=K, cam2w=cam2w, depthmaps=depth, pts3d=pts, losses=losses[:niter1 + niter2],
               # (dense unprojection reads _core with _cam_rows: both belong to the start of the last iteration, like the
               # depthmaps the reference unprojects from; the stepped core depths are params["core_depth"])
               _cam_rows=cam, _core=core_res, _base_focals=base_focals,
               _adam_m=work[:11 * Cn].clone(),  # first moments, order pps|log_focals|quats|trans|log_sizes (tests)
               _flags=work[22 * Cn + 24 * Cn + 20 * Cn:22 * Cn + 24 * Cn + 20 * Cn + 4].clone())  # loss, NaN stop
    if depth_work is not None:   # opt_depth: first moments of the core depths [C, G] (tests)
        n_rows2 = depth_work.numel() - 3 * Cn * G
        res["_adam_m_core"] = depth_work[n_rows2:n_rows2 + Cn * G].reshape(Cn, G).clone()
    params = dict(P)
    params["core_depth"] = core
    params["core_len"] = core_len          # true lengths of the padded rows (views of different sizes)
    res["core_len"] = core_len
    return res, params
