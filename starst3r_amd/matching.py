"""Reciprocal nearest-neighbour matching of dense descriptors (path A).

Mirrors Mast3r's `fast_reciprocal_NNs(pts1, pts2, subsample_or_initxy1=8, ret_xy=..., dist='dot',
block_size=2**13)` as the reference reaches it (starster/reconstruct.py:97 -> forward_mast3r ->
extract_correspondences): same seeds, same iteration, same convergence rule, same
unique/sort of the result.  The nearest-neighbour queries run on the MFMA kernel (st3r_nn_dot_argmax);
the iteration around them is device resident too (st3r_recip_nn); only the final unique/sort is torch.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops


def nn_dot_argmax(ctx, queries, db, want_score=False):
    """argmax_j queries[q] . db[j]  ->  int32 [n] (first index on ties)."""
    n, m, d = queries.shape[0], db.shape[0], db.shape[1]
    nn = torch.empty((n,), dtype=torch.int32, device=db.device)
    score = torch.empty((n,), dtype=torch.float32, device=db.device) if want_score else None
    if n:
        _lib.check(_lib.lib().st3r_nn_dot_argmax(ctx.handle, ops._stream(), ops._p(queries), n, ops._p(db), m, d,
                                                 ops._p(nn, torch.int32), ops._p(score)))
    return (nn, score) if want_score else nn


def merge_corres(idx1, idx2, shape1=None, shape2=None, ret_xy=True, ret_index=False):
    """unique correspondences sorted on (idx2, idx1) packed as one int64 -- Mast3r merge_corres; ret_index adds the
    position of the first occurrence of every kept pair in the input (numpy.unique(return_index=True))."""
    key = (idx2.to(torch.int64) & 0xFFFFFFFF) | (idx1.to(torch.int64) << 32)  # little endian view of np.c_[idx2, idx1]
    if ret_index:
        key, order = torch.sort(key, stable=True)
        first = torch.ones_like(key, dtype=torch.bool)
        first[1:] = key[1:] != key[:-1]
        key, index = key[first], order[first]
    else:
        key = torch.unique(key)  # sorted
    i2 = (key & 0xFFFFFFFF).to(torch.int32); i1 = (key >> 32).to(torch.int32)
    out = (i1, i2)
    if ret_xy and shape1 is not None:
        W1, W2 = shape1[1], shape2[1]
        out = (torch.stack((i1 % W1, torch.div(i1, W1, rounding_mode="floor")), dim=-1),
               torch.stack((i2 % W2, torch.div(i2, W2, rounding_mode="floor")), dim=-1))
    return out + (index,) if ret_index else out


def fast_reciprocal_NNs(pts1, pts2, subsample_or_initxy1=8, ret_xy=True, pixel_tol=0, ret_basin=False, device="cuda",
                        max_iter=10, **matcher_kw):
    """pts1 [H1,W1,D], pts2 [H2,W2,D] float32 descriptors -> matched (xy1, xy2) or flat indices (torch, on the device).
    Only the reference's configuration is implemented: integer subsample, pixel_tol=0, ret_basin=False, dist='dot'."""
    assert matcher_kw.get("dist", "dot") == "dot", "only dist='dot' (the reference's choice) is implemented"
    if pixel_tol != 0 or ret_basin or not isinstance(subsample_or_initxy1, (int, np.integer)):
        raise NotImplementedError("fast_reciprocal_NNs: only integer subsample, pixel_tol=0, ret_basin=False run on the HIP path")
    ctx = ops.get_context(device)
    dev = ctx.device
    H1, W1, D1 = pts1.shape; H2, W2, D2 = pts2.shape
    assert D1 == D2
    A = pts1.reshape(-1, D1).to(dev, torch.float32).contiguous()
    B = pts2.reshape(-1, D2).to(dev, torch.float32).contiguous()
    S = int(subsample_or_initxy1)
    lib = _lib.lib()
    n = lib.st3r_recip_nn_seed_count(H1, W1, S)
    xy1 = torch.empty((n,), dtype=torch.int32, device=dev); xy2 = torch.empty_like(xy1); notyet = torch.empty_like(xy1)
    if n:  # the whole reciprocal iteration runs on the device without a host round trip
        _lib.check(lib.st3r_recip_nn(ctx.handle, ops._stream(), ops._p(A), H1, W1, ops._p(B), H2, W2, D1, S, max_iter,
                                     ops._p(xy1, torch.int32), ops._p(xy2, torch.int32), ops._p(notyet, torch.int32)))
    converged = notyet == 0
    return merge_corres(xy1[converged], xy2[converged], (H1, W1), (H2, W2), ret_xy=ret_xy)


def install_into_mast3r():
    """Route Mast3r's own matching through the HIP kernels: replaces `fast_reciprocal_NNs` in `mast3r.fast_nn` and in
    the modules that imported it by name, with a wrapper that returns numpy arrays like upstream.  For users who have
    the `mast3r` package (the reference reaches it at starster/reconstruct.py:97); a no-op ImportError otherwise."""
    import importlib
    fast_nn = importlib.import_module("mast3r.fast_nn")

    def patched(pts1, pts2, subsample_or_initxy1=8, ret_xy=True, pixel_tol=0, ret_basin=False, device="cuda", **kw):
        a, b = fast_reciprocal_NNs(torch.as_tensor(pts1), torch.as_tensor(pts2), subsample_or_initxy1, ret_xy, pixel_tol,
                                   ret_basin, device, **kw)
        return a.cpu().numpy(), b.cpu().numpy()

    fast_nn.fast_reciprocal_NNs = patched
    for name in ("mast3r.cloud_opt.sparse_ga",):
        try:
            mod = importlib.import_module(name)
            if hasattr(mod, "fast_reciprocal_NNs"):
                mod.fast_reciprocal_NNs = patched
        except ImportError:
            pass
    return patched
