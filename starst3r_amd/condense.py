"""Canonical-data condensation between the matching (path A) and the alignment (path B) -- SURVEY 8(f) row 2.

Mirrors the Mast3r functions starster/reconstruct.py:101-106 calls (mast3r/cloud_opt/sparse_ga.py [U]; the submodule is
not vendored by the reference, so names, argument meaning and return structure follow the upstream source, and
oracle/condense_oracle.py is the pinned restatement of the arithmetic):

    prepare_canonical_data(imgs, tmp_pairs, subsample, ...) -> tmp_pairs, pairwise_scores, canonical_views,
                                                               canonical_paths, preds_21
    compute_min_spanning_tree(pairwise_scores)              -> (root, [(parent, child), ...])
    condense_data(imgs, tmp_pairs, canonical_views, preds_21) -> imsizes, pps, base_focals, core_depth, anchors,
                                                               corres, corres2d, preds_21

`tmp_pairs[(img1, img2)] = ((pred1, pred2), corres)` is what forward_mast3r caches per unordered pair: either the
tensors themselves or the paths of the `torch.save`d files (the reference's disk cache, starster/scene.py:122):
pred1 = (X11, C11, X21, C21) in the frame of img1, pred2 = (X22, C22, X12, C12) in the frame of img2,
corres = (score, (xy1, xy2, confs)).  The per-pixel arithmetic (canonical pointmaps, focals, anchor offsets) runs in
the HIP library (csrc/condense.hip); list plumbing stays on the host like upstream.  `condense()` chains the three
and returns the flat layout st3r_align_run consumes."""
import numpy as np
import torch

from . import ops


class Slice:
    """One ordered image pair of the reference's `imgs_slices` (reconstruct.py:280-290): the correspondences of the
    pair sit at anchors[img1][slice1] and anchors[img2][slice2]."""

    def __init__(self, img1, slice1, img2, slice2, confs):
        self.img1, self.slice1, self.img2, self.slice2, self.confs = img1, slice1, img2, slice2, confs



def _load(x, device):
    if isinstance(x, (str, bytes)) or hasattr(x, "__fspath__"):
        x = torch.load(x, map_location=device)
    return x


def _dev(x, device, dtype=torch.float32):
    return torch.as_tensor(x).to(device=device, dtype=dtype).contiguous()


def canonical_view(ptmaps11, confs11, subsample, mode="avg-angle", ctx=None):
    """-> (canon [H,W,4] = canonical pointmap + relative depth map, confs [H,W]) like upstream."""
    assert mode == "avg-angle", "the reference calls prepare_canonical_data(mode='avg-angle') (reconstruct.py:102)"
    ctx = ctx or ops.get_context(str(confs11.device))
    canon, canon2, cconf = ops.canon_view(ctx, ptmaps11.contiguous(), confs11.contiguous(), subsample)
    return torch.cat([canon, canon2[..., None]], dim=-1), cconf


def prepare_canonical_data(imgs, tmp_pairs, subsample, order_imgs=False, min_conf_thr=0, cache_path=None,
                           device="cuda:0", mode="avg-angle", canon_out=None, **kw):
    """`canon_out` (optional dict) receives {img: (canon2 [H,W], cconf [H,W])}: what upstream keeps in its canonical
    cache files for SparseGA.get_dense_pts3d."""
    ctx = ops.get_context(str(device))
    C = len(imgs)
    pairwise_scores = torch.zeros((C, C), device=device)
    canonical_views, preds_21, canonical_paths = {}, {}, []
    for img in imgs:
        ptmaps11, confs11, pixels = [], [], {}
        for (img1, img2), ((path1, path2), path_corres) in tmp_pairs.items():
            if img not in (img1, img2):
                continue
            first = img == img1
            X, Cf, X2, C2 = [_dev(t, device) for t in _load(path1 if first else path2, device)]
            score, (xy1, xy2, confs) = _load(path_corres, device)
            xy, confs = _dev(xy1 if first else xy2, device), _dev(confs, device)
            if min_conf_thr:
                keep = confs >= min_conf_thr
                xy, confs = xy[keep], confs[keep]
            other = img2 if first else img1
            pixels[other] = (xy, confs)
            preds_21.setdefault(img, {})[other] = (X2[::subsample, ::subsample].reshape(-1, 3).contiguous(),
                                                   C2[::subsample, ::subsample].reshape(-1).contiguous())
            i, j = imgs.index(img1), imgs.index(img2)
            pairwise_scores[i, j] = pairwise_scores[j, i] = float(score[2])   # number of correspondences [U]
            ptmaps11.append(X); confs11.append(Cf)
        if not ptmaps11:
            raise ValueError(f"image {img!r} appears in no pair")
        canon4, cconf = canonical_view(torch.stack(ptmaps11), torch.stack(confs11), subsample, mode, ctx=ctx)
        H, W = cconf.shape
        pp = torch.tensor([W / 2, H / 2], device=device)
        core_depth = canon4[subsample // 2::subsample, subsample // 2::subsample, 2].contiguous()
        canon2 = canon4[..., 3].contiguous()
        idxs, offsets = {}, {}
        for other, (xy, _c) in pixels.items():
            idx, off = ops.anchor_offsets(ctx, canon2, xy, subsample)
            idxs[other], offsets[other] = idx.long(), off
        canonical_views[img] = [pp, (H, W), canon4[..., :3], core_depth, pixels, idxs, offsets]   # focal: below
        if canon_out is not None:
            canon_out[img] = (canon2, cconf)
    # focals of all images: one launch per image size (st3r_focal_weiszfeld_batch), not one per image
    by_size = {}
    for img in imgs:
        by_size.setdefault(canonical_views[img][1], []).append(img)
    for (H, W), group in by_size.items():
        focals = ops.focal_weiszfeld_batch(ctx, torch.stack([canonical_views[i][2] for i in group]).contiguous(),
                                           (W / 2, H / 2), 0.5, 3.5)
        for k, i in enumerate(group):
            canonical_views[i][2] = focals[k:k + 1].view(1)
    canonical_views = {k: tuple(v) for k, v in canonical_views.items()}
    return tmp_pairs, pairwise_scores, canonical_views, canonical_paths, preds_21


def compute_min_spanning_tree(pws):
    """Maximum-score spanning tree (Kruskal on the host, like upstream's scipy call), rooted at the end of the best
    edge with the larger total score, edges in breadth-first order as (parent, child)."""
    pws = pws.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(pws) else np.asarray(pws, np.float64)
    C = pws.shape[0]
    edges = sorted(((-pws[i, j], i, j) for i in range(C) for j in range(i + 1, C) if pws[i, j] > 0))
    if not edges:
        return 0, []
    comp = list(range(C))

    def find(a):
        while comp[a] != a:
            comp[a] = comp[comp[a]]; a = comp[a]
        return a
    adj = {i: [] for i in range(C)}
    for _, i, j in edges:
        ri, rj = find(i), find(j)
        if ri != rj:
            comp[ri] = rj; adj[i].append(j); adj[j].append(i)
    _, bi, bj = edges[0]
    root = bi if pws[bi].sum() >= pws[bj].sum() else bj
    seen, order, out = {root}, [root], []
    for a in order:
        for b in sorted(adj[a]):
            if b not in seen:
                seen.add(b); order.append(b); out.append((a, b))
    return root, out


def condense_data(imgs, tmp_pairs, canonical_views, preds_21, dtype=torch.float32):
    set_imgs = set(imgs)
    pps, shapes, focals, core_depth, img_anchors, tmp_pixels = [], [], [], [], {}, {}
    for idx1, img1 in enumerate(imgs):
        pp, shape, focal, cdepth, pixels_confs, idxs, offsets = canonical_views[img1]
        pps.append(pp); shapes.append(shape); focals.append(focal); core_depth.append(cdepth)
        uv, ii, oo, cur = [], [], [], 0
        for img2, (pixels, match_confs) in pixels_confs.items():
            if img2 not in set_imgs:
                continue
            assert len(pixels) == len(idxs[img2]) == len(offsets[img2])
            uv.append(pixels); ii.append(idxs[img2]); oo.append(offsets[img2])
            tmp_pixels[img1, img2] = (pixels.to(dtype), match_confs.to(dtype), slice(cur, cur + len(pixels)))
            cur += len(pixels)
        img_anchors[idx1] = (torch.cat(uv), torch.cat(ii), torch.cat(oo))
    imgs_slices, all_confs = [], []
    corres2d = {i: [] for i in range(len(imgs))}
    for img1, img2 in tmp_pairs:
        if (img1, img2) not in tmp_pixels or (img2, img1) not in tmp_pixels:
            continue
        pix1, confs1, slice1 = tmp_pixels[img1, img2]
        pix2, confs2, slice2 = tmp_pixels[img2, img1]
        i1, i2 = imgs.index(img1), imgs.index(img2)
        confs = (confs1 * confs2).sqrt()
        all_confs.append(confs)
        # both orders: the reference looks is_matching_ok up as [img1, img2] and [img2, img1] (reconstruct.py:286,299)
        imgs_slices.append(Slice(i1, slice1, i2, slice2, confs))
        imgs_slices.append(Slice(i2, slice2, i1, slice1, confs))
        corres2d[i1].append((pix1, confs, i2, slice2)); corres2d[i2].append((pix2, confs, i1, slice1))
    all_confs = torch.cat(all_confs) if all_confs else torch.zeros(0)
    corres = (all_confs, float(all_confs.sum()), imgs_slices)

    def aggreg(img1, lst):
        if not lst:
            z = torch.zeros(0, device=pps[0].device)
            return img1, z.reshape(0, 2), z, 0.0, []
        pix1, confs, img2, slice2 = zip(*lst)
        cf = torch.cat(confs).to(dtype)
        return img1, torch.cat(pix1).to(dtype), cf, float(cf.sum()), list(zip(img2, slice2))
    corres2d = [aggreg(i, lst) for i, lst in corres2d.items()]
    imsizes = torch.tensor([(W, H) for H, W in shapes], device=pps[0].device)
    # regression-fallback targets follow img1's anchors (upstream: pred[idxs of the anchors of img1])
    sub = {}
    for k2, d in preds_21.items():
        sub[k2] = {}
        for k1, (pred, conf) in d.items():
            idx = img_anchors[imgs.index(k1)][1]
            sub[k2][k1] = (pred[idx], conf[idx])
    return imsizes, torch.stack(pps), torch.cat(focals), core_depth, img_anchors, corres, corres2d, sub


def dense_table(imgs, canonical_views, canon_maps, subsample, device="cuda:0"):
    """Every pixel of every view as an anchor of the view's core depthmap (SparseGA.get_dense_pts3d [U] does the
    same with anchor_depth_offsets on the full pixel grid): list of dict(pixels, idxs, offsets, confs, base_focal)
    in the order of `imgs` -- the table SparseGAResult.get_dense_pts3d unprojects."""
    ctx = ops.get_context(str(device))
    out = []
    for img in imgs:
        _pp, (H, W), focal, _core, _pix, _i, _o = canonical_views[img]
        canon2, cconf = canon_maps[img]
        ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        xy = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).float().contiguous()
        idx, off = ops.anchor_offsets(ctx, canon2, xy, subsample)
        out.append(dict(pixels=xy, idxs=idx, offsets=off, confs=cconf.reshape(-1).contiguous(),
                        base_focal=float(focal[0])))
    return out


def condense(imgs, tmp_pairs, subsample=8, device="cuda:0", matching_conf_thr=5.0, with_dense=False):
    """forward_mast3r's per-pair cache -> the flat alignment problem (synth_align.flatten layout) for align.run;
    with_dense adds the "dense" pixel table of reconstruct.SparseGAResult."""
    from .reconstruct import flatten_reference_inputs
    canon_maps = {} if with_dense else None
    tmp_pairs, scores, views, _paths, preds_21 = prepare_canonical_data(imgs, tmp_pairs, subsample, device=device,
                                                                       canon_out=canon_maps)
    mst = compute_min_spanning_tree(scores)
    imsizes, pps, base_focals, core_depth, anchors, corres, corres2d, preds = condense_data(imgs, tmp_pairs, views,
                                                                                             preds_21)
    flat = flatten_reference_inputs(imgs, imsizes, pps, base_focals, core_depth, anchors, corres, corres2d, preds, mst,
                                    matching_conf_thr=matching_conf_thr)
    flat["subsample"] = np.int64(subsample)
    if with_dense:
        flat["dense"] = dense_table(imgs, views, canon_maps, subsample, device)
    return flat
