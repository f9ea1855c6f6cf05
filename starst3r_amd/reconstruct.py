"""Reconstruction pipeline -- drop-in for the reference's `starster.reconstruct`
(starster/reconstruct.py:19-113): `reconstruct_scene(model, imgs, filelist, device, optim_params, tmpdir)`
returns the TUPLE (scene, optim_params) exactly like the reference (its docstring says "scene", the code and
its only caller use a tuple: reconstruct.py:72,113; scene.py:122).

What runs where:
  * pairwise inference (the Mast3r ViT) stays on PyTorch-ROCm and is supplied by `model`;
  * reciprocal-NN matching (path A) -> starst3r_amd.matching (MFMA kernel);
  * global alignment (path B)       -> starst3r_amd.align   (fused HIP optimiser);
  * the condensation between them (canonical pointmaps, focals, MST, anchors; Mast3r prepare_canonical_data /
    compute_min_spanning_tree / condense_data) -> starst3r_amd.condense (HIP kernels + host lists).

Model protocol.  `model` is a Mast3r network like in the reference (its ViT forward then runs through Mast3r's own
`symmetric_inference`, wrapped by starst3r_amd.forward.Mast3rNetwork; the package is not vendored by the reference --
empty submodule -- and must be installed for that), or any object with one of
      model.symmetric_inference(img1, img2, device) -> (res11, res21, res22, res12)
the network alone (head outputs 'pts3d', 'conf', 'desc', 'desc_conf' per pair, like Mast3r's symmetric_inference):
pair list, reciprocal matching, the resumable disk cache (starst3r_amd.forward), condensation and alignment run here;
      model.forward_pairs(imgs, filelist, device, cache_dir) -> (tmp_pairs, images)
the per-pair cache of Mast3r's forward_mast3r (see starst3r_amd.condense for the layout; tensors or torch.save
paths) plus the resized images -- condensation and alignment then run here --, or
      model.condense(imgs, filelist, device, cache_dir) -> dict
returning the already condensed problem in st3r_synth.synth_align.flatten() layout plus
      "imgs": list of HxWx3 float arrays in [0,1] (the Mast3r-resized GT images),
      "dense": optional per-view dict(pixels [n,2], idxs [n], offsets [n], confs [n], base_focal) -- the dense
               pixels as anchors of the view's core depthmap (dense unprojection).
st3r_synth.synth_model.SyntheticPairwiseModel implements it on synthetic scenes (BASELINE configs[0]).
"""
__all__ = ("reconstruct_scene", "reconstruct", "run_sparse_ga", "sparse_scene_optimizer_slam",
           "flatten_reference_inputs")

import math
import tempfile

import numpy as np
import torch

from . import align


class SparseGAResult:
    """The members of Mast3r's SparseGA that the reference touches (scene.py:133,138-139,148):
    .imgs, .cam2w, .intrinsics, .get_dense_pts3d(clean_depth=True) -> (pts list, depthmaps list, confs list)."""

    def __init__(self, imgs, res, dense=None):
        self.imgs = imgs
        self.cam2w = res["cam2w"]
        self.intrinsics = res["intrinsics"]
        self.depthmaps = res["depthmaps"]
        self.pts3d = res["pts3d"]
        self.losses = res["losses"]
        self._res = res
        self._dense = dense

    def get_dense_pts3d(self, clean_depth=True):
        """Dense unprojection with the OPTIMISED cameras and depthmaps (Mast3r SparseGA.get_dense_pts3d [U]): every
        dense pixel is treated like an anchor -- depth = depthmap[idx] * offset' -- so the points live in the
        optimiser's gauge (App. A.5 make_pts3d); with clean_depth the confidences of points floating in front of
        another view's surface are lowered (dust3r clean_pointcloud [U]).  Runs in libst3r_hip.so
        (st3r_dense_unproject / st3r_dense_clean).  Returns (pts list, depthmaps list, confs list) per view."""
        if self._dense is None:
            raise NotImplementedError("dense unprojection needs the model's dense pixel table (SURVEY.md 8(f) #3)")
        from . import ops
        dev = self.cam2w.device
        ctx = ops.get_context(dev)
        counts = [len(d["idxs"]) for d in self._dense]
        start = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=dev)
        def cat(key, dt):   # numpy arrays or tensors (host or device), one entry per view
            return torch.cat([(d[key] if torch.is_tensor(d[key]) else torch.as_tensor(np.asarray(d[key]))).to(
                device=dev, dtype=dt) for d in self._dense]).contiguous()
        pix, idx, off, conf = cat("pixels", torch.float32), cat("idxs", torch.int32), cat("offsets", torch.float32), \
            cat("confs", torch.float32)
        pts, z = ops.dense_unproject(ctx, start, pix, idx, off, self._res["_core"], self._res["_cam_rows"],
                                     self._res["_base_focals"])
        if clean_depth:
            sizes = torch.tensor([list(np.asarray(im).shape[:2]) for im in self.imgs], dtype=torch.int32, device=dev)
            assert all(int(h) * int(w) == c for (h, w), c in zip(sizes.tolist(), counts)), \
                "clean_depth needs one dense entry per pixel, in raster order"
            conf = ops.dense_clean(ctx, start, sizes, self._res["_cam_rows"], pts, z, conf)
        bounds = start.tolist()
        sl = [slice(bounds[i], bounds[i + 1]) for i in range(len(counts))]
        return [pts[s_] for s_ in sl], list(self.depthmaps), [conf[s_].cpu() for s_ in sl]


def flatten_reference_inputs(imgs, imsizes, pps, base_focals, core_depth, anchors, corres, corres2d, preds_21, mst,
                             matching_conf_thr=5.0):
    """The set-up block of the reference optimiser (starster/reconstruct.py:148-207, 263-309) restated as array
    plumbing: takes the SAME objects `sparse_scene_optimizer_slam` receives from Mast3r's condense_data and returns
    the flat layout the C ABI consumes (st3r_synth.synth_align.flatten documents it).

      anchors   {img index: (pixels [n,2], idxs [n], offsets [n])}
      corres    (_, _, imgs_slices) with .img1 .slice1 .img2 .slice2 .confs per ORDERED pair
      corres2d  [(img1, pix1 [m,2], confs [m], confsum, [(img2, slice2), ...]), ...]
      preds_21  {name of img2: {name of img1: (pts [k,3] in cam2, conf [k])}}  (regression fallback)
      mst       (root, [(i, j), ...])

    A pair passes the matching gate when `confs.max() > matching_conf_thr` (:283); its rows feed loss_3d / loss_2d
    (:325-369), the others the DUSt3R regression (:311-323)."""
    def npy(x, dt):
        x = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
        return np.ascontiguousarray(x).astype(dt)

    C = len(imgs)
    # the reference (and Mast3r) keep per-view lists and accept mixed image sizes (reconstruct.py:170-177, 276);
    # st3r_align_run takes one [C, G] block: the rows are padded to the longest view, `core_len` keeps the true lengths
    # (an anchor only ever indexes its own view's part)
    rows = [npy(d, np.float32).reshape(-1) for d in core_depth]
    core_len = np.array([len(r) for r in rows], np.int64)
    core_pad = np.ones((C, int(core_len.max())), np.float32)
    for v, r in enumerate(rows):
        core_pad[v, :len(r)] = r
    counts = [len(anchors[v][1]) for v in range(C)]
    anchor_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rng = lambda sl, n: np.arange(n)[sl]
    out = dict(n_views=np.int64(C), imsizes=npy(imsizes, np.int64), pps=npy(torch.stack(list(pps)) if isinstance(pps, (list, tuple)) else pps, np.float32),
               base_focals=npy(torch.stack([torch.as_tensor(f).reshape(()) for f in base_focals]) if isinstance(base_focals, (list, tuple)) else base_focals, np.float32).reshape(-1),
               core_depth=core_pad, core_len=core_len, anchor_off=anchor_off,
               anchor_pix=np.concatenate([npy(anchors[v][0], np.float32) for v in range(C)]),
               anchor_idx=np.concatenate([npy(anchors[v][1], np.int64) for v in range(C)]),
               anchor_offset=np.concatenate([npy(anchors[v][2], np.float32) for v in range(C)]),
               anchor_img=np.concatenate([np.full(counts[v], v, np.int32) for v in range(C)]),
               mst_root=np.int64(mst[0]), mst_edges=np.array(mst[1], np.int64).reshape(-1, 2))
    _, _, imgs_slices = corres
    ok = {(s.img1, s.img2): bool(npy(s.confs, np.float32).max() > matching_conf_thr) for s in imgs_slices}
    a1, a2, cf = [], [], []
    d_a1, d_tgt, d_img2, d_conf = [], [], [], []
    for s in imgs_slices:
        if ok[s.img1, s.img2]:                                   # loss3d_slices (:290)
            a1.append(anchor_off[s.img1] + rng(s.slice1, counts[s.img1]))
            a2.append(anchor_off[s.img2] + rng(s.slice2, counts[s.img2]))
            cf.append(npy(s.confs, np.float32))
        else:                                                    # dust3r_slices (:289, 315-322)
            tgt, tc = preds_21[imgs[s.img2]][imgs[s.img1]]
            tc = npy(tc, np.float32)
            d_a1.append(anchor_off[s.img1] + np.arange(len(tc))); d_tgt.append(npy(tgt, np.float32))
            d_img2.append(np.full(len(tc), s.img2, np.int32)); d_conf.append(tc)
    c_pix, c_a2, c_conf, c_img1 = [], [], [], []
    for (img1, pix1, confs, _confsum, slices) in corres2d:      # cleaned_corres2d (:291-309)
        pix1 = npy(pix1, np.float32); confs = npy(confs, np.float32)
        cur = 0
        for img2, slice2 in slices:
            idx2 = rng(slice2, counts[img2])
            n = len(idx2)
            if ok[img1, img2]:
                c_pix.append(pix1[cur:cur + n]); c_conf.append(confs[cur:cur + n]); c_a2.append(anchor_off[img2] + idx2)
                c_img1.append(np.full(n, img1, np.int32))
            cur += n
    cat = lambda xs, dt, shape=(0,): np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dt)
    out.update(corr_a1=cat(a1, np.int64), corr_a2=cat(a2, np.int64), corr_conf=cat(cf, np.float32),
               c2d_pix=cat(c_pix, np.float32, (0, 2)), c2d_a2=cat(c_a2, np.int64), c2d_conf=cat(c_conf, np.float32),
               c2d_img1=cat(c_img1, np.int32), dust_a1=cat(d_a1, np.int64), dust_tgt=cat(d_tgt, np.float32, (0, 3)),
               dust_img2=cat(d_img2, np.int32), dust_conf=cat(d_conf, np.float32))
    return out


def l1_loss(x, y):
    """Euclidean distance of two point sets (the `l1_loss` Mast3r's gamma_loss builds on)."""
    return torch.linalg.norm(x - y, dim=-1)


def gamma_loss(gamma):
    """The robust loss the reference passes as loss1 / loss2 / lossd (starster/reconstruct.py:118-120):
    rho(x, y) = (|x - y| + off)^gamma - off^gamma, off = (1/gamma)^(1/(gamma-1)); gamma = 1 is `l1_loss`.  The returned
    callable carries `.gamma`, which is all the library needs (the loss itself is evaluated inside k_align_resid)."""
    if gamma == 1:
        return l1_loss
    off = (1 / gamma) ** (1 / (gamma - 1))

    def loss_func(x, y):
        return (l1_loss(x, y) + off) ** gamma - off ** gamma
    loss_func.gamma = float(gamma)
    return loss_func


def cosine_schedule(alpha, lr_base, lr_end=0.0):
    return lr_end + (lr_base - lr_end) * (1 + math.cos(alpha * math.pi)) / 2


def linear_schedule(alpha, lr_base, lr_end=0.0):
    return (1 - alpha) * lr_base + alpha * lr_end


def _gamma_of(loss, default):
    """Exponent of a gamma_loss object: ours (`.gamma`), Mast3r's closure (free variables gamma / mul / offset / clip), or
    l1_loss (gamma = 1).  Anything else cannot be evaluated by the kernels."""
    if loss is None:
        return default
    g = getattr(loss, "gamma", None)
    if g is None and getattr(loss, "__name__", "") == "l1_loss":
        g = 1.0
    if g is None and getattr(loss, "__closure__", None):
        fv = dict(zip(loss.__code__.co_freevars, (c.cell_contents for c in loss.__closure__)))
        if "gamma" in fv:
            g = float(fv["gamma"])
            off = fv.get("offset")
            # the offset gamma_loss itself derives: (1/g)^(1/(g-1)); gamma = 1 is the plain distance (no offset)
            want = 0.0 if g == 1 else ((1 / g) ** (1 / (g - 1)) if g > 0 else None)
            if fv.get("mul", 1) != 1 or fv.get("clip", float("inf")) != float("inf") or \
                    (off is not None and (want is None or abs(off - want) > 1e-6 * max(abs(off), 1e-30))):
                raise NotImplementedError("gamma_loss with mul / clip / a custom offset is not implemented on the HIP path")
    # Mast3r's meta_gamma_loss (a loss factory the optimiser calls with the progress alpha, reconstruct.py:375,387) is
    # recognised by its name, not by a substring of repr(): module paths or closures may well contain "meta"
    name = getattr(loss, "__qualname__", getattr(loss, "__name__", type(loss).__name__))
    if g is None or "meta_gamma_loss" in name or type(loss).__name__ == "meta_gamma_loss":
        raise NotImplementedError("only gamma_loss(g) / l1_loss objects can be evaluated on the HIP path")
    if not g > 0:
        raise ValueError("gamma_loss: gamma must be positive")
    return float(g)


def sparse_scene_optimizer_slam(imgs, subsample, imsizes, pps, base_focals, core_depth, anchors, corres, corres2d,
                                preds_21, canonical_paths, mst, cache_path=None, lr1=0.2, niter1=500, loss1=None,
                                lr2=0.02, niter2=500, loss2=None, lossd=None, opt_pp=True, opt_depth=True,
                                schedule=None, depth_mode="add", exp_depth=False, lora_depth=False,
                                shared_intrinsics=False, init=None, device="cuda", dtype=torch.float32,
                                matching_conf_thr=5.0, loss_dust3r_w=0.01, verbose=True, dbg=(), prev_params=None):
    """Same signature and return value as the reference's optimiser (starster/reconstruct.py:116-457):
    `(imgs, res_coarse, res_fine, params_ret)` with `res = dict(intrinsics, cam2w, depthmaps, pts3d)` and
    `params_ret` the dict of per-view parameter lists a later call accepts as `prev_params`.  The optimisation
    itself is st3r_align_run_opts.  Implemented: the configuration the reference uses (:61-69) and the options that
    only change constants of the loop -- loss1 / loss2 / lossd = gamma_loss(g) (any g; `gamma_loss` below, Mast3r's own
    closures, or `l1_loss`), any `schedule(alpha, lr_base, lr_end)` callable, `opt_pp`, `opt_depth`.  shared_intrinsics,
    exp_depth, lora_depth, depth_mode != 'add' and `init` raise.  (shared_intrinsics is not just unimplemented: the
    reference hands the ONE shared pp / log_focal Parameter to Adam once per view (:193-197, :373), so what a step does
    to it -- one update or C, with the step count advancing by C -- depends on which torch.optim implementation (for-loop
    or foreach) the installed torch picks; there is no single behaviour to reproduce.)  The stage-1 result is not kept separately (the
    reference's only caller takes `res_fine or res_coarse`, :113): `res_coarse` is `res_fine` when a second stage ran."""
    if shared_intrinsics or exp_depth or lora_depth or depth_mode != "add" or init:
        raise NotImplementedError("shared_intrinsics, exp_depth, lora_depth, depth_mode != 'add' and per-image `init` are "
                                  "not implemented on the HIP path (the reference's own call uses none of them)")
    g1, g2, gd = _gamma_of(loss1, 1.1), _gamma_of(loss2, 0.4), _gamma_of(lossd, 1.1)
    flat = flatten_reference_inputs(imgs, imsizes, pps, base_focals, core_depth, anchors, corres, corres2d, preds_21,
                                    mst, matching_conf_thr)
    dev = "cuda:0" if str(device) == "cuda" else str(device)
    res, params = align.run(flat, lr1=lr1, niter1=niter1, lr2=lr2, niter2=niter2, prev_params=prev_params,
                            loss_dust3r_w=loss_dust3r_w, device=dev, schedule=schedule, gamma1=g1, gamma2=g2, gammad=gd,
                            opt_pp=bool(opt_pp), opt_depth=bool(opt_depth))
    off = flat["anchor_off"]
    C = len(imgs)
    clen = flat["core_len"]
    out = dict(intrinsics=res["intrinsics"], cam2w=res["cam2w"],
               depthmaps=[res["depthmaps"][v][:int(clen[v])] for v in range(C)],
               pts3d=[res["pts3d"][int(off[v]):int(off[v + 1])] for v in range(C)], losses=res["losses"], _res=res)
    params_ret = {k: [params[k][v] for v in range(C)] for k in ("pps", "log_focals", "quats", "trans", "log_sizes")}
    params_ret["core_depth"] = [params["core_depth"][v][:int(clen[v])] for v in range(C)]
    return imgs, out, (out if niter2 else None), params_ret


def _images_of_pairs(filelist, pairs):
    """The resized GT images, HxWx3 in [0,1], per file of `filelist` -- from the (3,H,W) tensors in [-1,1] the pair
    dicts carry (Mast3r's SparseGA keeps exactly these as `.imgs`, scene.py:133)."""
    found = {}
    for pair in pairs:
        for v in pair:
            if v["instance"] not in found:
                im = v["img"][0] if v["img"].dim() == 4 else v["img"]
                found[v["instance"]] = ((im.detach().float().cpu().permute(1, 2, 0) + 1) / 2).clamp(0, 1).numpy()
    return [found[n] for n in filelist]


def run_sparse_ga(imgs, pairs_in=None, cache_path=None, model=None, subsample=8, desc_conf="desc_conf", device="cuda",
                  dtype=torch.float32, shared_intrinsics=False, optim_params=None, **kw):
    """Reference signature (starster/reconstruct.py:75-113): returns (scene, optim_params).

    Two ways in:
      * `imgs` is the condensed dict of the `model.condense()` protocol (module docstring): aligned directly;
      * `imgs` is the reference's file list, `pairs_in` the list of (view dict, view dict) pairs and `model` a network:
        the reference's own sequence (:95-113) -- pair naming, forward_mast3r (pairwise inference through the network,
        reciprocal matching = path A, resumable disk cache), prepare_canonical_data / compute_min_spanning_tree /
        condense_data, the alignment (path B), and a scene object with the members of Mast3r's SparseGA
        the reference touches -- every step in this library (starst3r_amd.forward / condense / align).  A Mast3r network
        (the reference's model type) is wrapped in forward.Mast3rNetwork: only the ViT forward is Mast3r's."""
    kw.setdefault("lr1", 0.07); kw.setdefault("niter1", 500); kw.setdefault("lr2", 0.014); kw.setdefault("niter2", 200)
    dev = "cuda:0" if str(device) == "cuda" else str(device)
    if isinstance(imgs, dict):
        condensed = imgs
    else:
        if shared_intrinsics:
            raise NotImplementedError("shared_intrinsics is not implemented on the HIP path (see sparse_scene_optimizer_slam)")
        if kw.get("opt_depth"):
            raise NotImplementedError("run_sparse_ga aligns with opt_depth=False like the reference's call (:66)")
        from . import condense as _condense
        from .forward import forward_mast3r, wrap_network
        filelist = list(imgs)
        pairs = list(pairs_in)
        for pair in pairs:                                           # convert_dust3r_pairs_naming (:95)
            for v in pair:
                v["instance"] = filelist[v["idx"]]
        tmp_pairs, cache_path = forward_mast3r(pairs, wrap_network(model), cache_path, desc_conf=desc_conf, device=dev,
                                               subsample=subsample)
        condensed = _condense.condense(filelist, tmp_pairs, subsample, device=dev, with_dense=True,
                                       matching_conf_thr=float(kw.get("matching_conf_thr", 5.0)))
        condensed["imgs"] = _images_of_pairs(filelist, pairs)
    res, params = align.run(condensed, lr1=kw["lr1"], niter1=kw["niter1"], lr2=kw["lr2"], niter2=kw["niter2"],
                            prev_params=optim_params, device=dev)
    return SparseGAResult(condensed.get("imgs"), res, condensed.get("dense")), params


def reconstruct_scene(model, imgs, filelist, device, optim_params=None, tmpdir=None):
    """Run the reconstruction pipeline: pairwise inference + matching (`model`), then global alignment.

    Returns (scene, optim_params); pass optim_params back in to warm start after adding images
    (starster/reconstruct.py:19-72).  `model`: a Mast3r network like in the reference, or any object implementing one
    of the protocols of the module docstring."""
    if tmpdir is None:
        tmpdir = tempfile.mkdtemp()
    if hasattr(model, "forward_pairs") and not hasattr(model, "symmetric_inference"):
        from . import condense as _condense
        tmp_pairs, images = model.forward_pairs(imgs, filelist, device, tmpdir)
        condensed = _condense.condense(list(filelist), tmp_pairs, getattr(model, "subsample", 8), device=device,
                                       with_dense=True)
        condensed["imgs"] = images
        return run_sparse_ga(condensed, device=device, optim_params=optim_params)
    if hasattr(model, "condense") and not hasattr(model, "symmetric_inference"):
        condensed = model.condense(imgs, filelist, device, tmpdir)
        return run_sparse_ga(condensed, device=device, optim_params=optim_params)
    # the network only -- an object with symmetric_inference, or a Mast3r network (wrapped by run_sparse_ga): the
    # reference's own sequence (reconstruct.py:51-69): views, the complete symmetrized pair graph, run_sparse_ga
    from .image import make_pair_indices, prepare_images_for_mast3r
    views = prepare_images_for_mast3r(imgs)
    pairs = [(views[i], views[j]) for i, j in make_pair_indices(len(views), symmetric=True)]
    return run_sparse_ga(filelist, pairs, tmpdir, model, subsample=getattr(model, "subsample", 8), device=device,
                         optim_params=optim_params, lr1=0.07, niter1=500, lr2=0.014, niter2=200, opt_depth=False,
                         matching_conf_thr=5, shared_intrinsics=False)


reconstruct = reconstruct_scene  # alias for the wording of BASELINE.json's north_star
