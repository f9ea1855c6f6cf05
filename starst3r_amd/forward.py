"""Pairwise inference bookkeeping, matching and the on-disk pair cache -- what the reference gets from Mast3r's
`forward_mast3r` at starster/reconstruct.py:97 (mast3r/cloud_opt/sparse_ga.py [U], not vendored): for every unordered
image pair the network runs symmetrically (the ViT itself stays the model's business), the four descriptor maps are
matched with reciprocal nearest neighbours (path A: st3r_recip_nn, MFMA kernel) and everything is `torch.save`d under
`cache_path` with the upstream file layout

    forward/<md5 img1>/<md5 img2>.pth                 (X11, C11, X21, C21)   pointmaps / confidences, frame of img1
    forward/<md5 img2>/<md5 img1>.pth                 (X22, C22, X12, C12)
    corres_conf=<desc_conf>_subsample=<s>/<md5 1>-<md5 2>.pth   ((conf score, sum of confs, count), (xy1, xy2, confs))

so that a second call with more images (Scene.add_images, starster/scene.py:117-122: the file names are the fake
"0.png", "1.png", ...) only infers the new pairs.  SURVEY 8(f) row 4 (pair cache) and the caller side of path A.

Model protocol: `model.symmetric_inference(img1, img2, device)` -> (res11, res21, res22, res12), dicts with
'pts3d' [1,H,W,3], 'conf' [1,H,W], 'desc' [1,H,W,D], 'desc_conf' [1,H,W] like Mast3r's heads.  A Mast3r network itself
(mast3r.model.AsymmetricMASt3R, what the reference hands over: main.py:46, starster/__init__.py:3) has no such method --
upstream it is the module-level function mast3r.cloud_opt.sparse_ga.symmetric_inference(model, img1, img2, device) [U]
-- so `wrap_network` puts the `Mast3rNetwork` adaptor around any object without the protocol: the ViT runs through
Mast3r's own function, everything behind it (matching, cache, condensation, alignment, dense points) in this library."""
import hashlib
import os

import torch

from . import matching


class Mast3rNetwork:
    """Adaptor for the reference's own model type.  `model` is the network object of the reference's call
    `reconstruct_scene(model, ...)` (starster/reconstruct.py:19,95-99: an AsymmetricMASt3R instance); the head outputs
    of a pair come from Mast3r's `symmetric_inference(model, img1, img2, device)` [U], imported when first needed (the
    package is not vendored by the reference -- empty submodule -- so its absence is reported then, loudly)."""

    def __init__(self, model, subsample=8):
        self.model = model
        self.subsample = subsample
        self._fn = None

    def symmetric_inference(self, img1, img2, device):
        if self._fn is None:
            try:
                from mast3r.cloud_opt.sparse_ga import symmetric_inference
            except ImportError as e:
                raise ImportError(
                    "a Mast3r network needs the `mast3r` package for its pairwise inference "
                    "(mast3r.cloud_opt.sparse_ga.symmetric_inference); it is not vendored by the reference (empty "
                    "submodule).  Without it pass a model that implements symmetric_inference(img1, img2, device), "
                    "forward_pairs(...) or condense(...) (starst3r_amd.reconstruct)") from e
            self._fn = symmetric_inference
        return self._fn(self.model, img1, img2, device)


def wrap_network(model):
    """`model` itself when it implements one of the model protocols of starst3r_amd.reconstruct, else the Mast3rNetwork
    adaptor around it (a Mast3r network: the reference's own model type)."""
    if any(hasattr(model, a) for a in ("symmetric_inference", "forward_pairs", "condense")):
        return model
    return Mast3rNetwork(model, getattr(model, "subsample", 8))


def hash_md5(s):
    return hashlib.md5(s.encode("utf-8")).hexdigest()


def _mkdir_for(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    return path


def extract_correspondences(feats, qonfs, subsample=8, device="cuda:0"):
    """Mast3r extract_correspondences [U]: reciprocal matches of (desc11, desc21) and of (desc12, desc22), both
    directions each, merged; confidence = sqrt(qonf1 qonf2) at the matched pixels."""
    feat11, feat21, feat22, feat12 = feats
    qonf11, qonf21, qonf22, qonf12 = qonfs
    assert feat11.shape[:2] == feat12.shape[:2] == qonf11.shape == qonf12.shape
    assert feat21.shape[:2] == feat22.shape[:2] == qonf21.shape == qonf22.shape
    idx1, idx2, q1, q2 = [], [], [], []
    for A, B, QA, QB in ((feat11, feat21, qonf11, qonf21), (feat12, feat22, qonf12, qonf22)):
        a12, b12 = matching.fast_reciprocal_NNs(A, B, subsample_or_initxy1=subsample, ret_xy=False, device=device)
        b21, a21 = matching.fast_reciprocal_NNs(B, A, subsample_or_initxy1=subsample, ret_xy=False, device=device)
        i1 = torch.cat([a12, a21]).long(); i2 = torch.cat([b12, b21]).long()
        idx1.append(i1); idx2.append(i2)
        q1.append(QA.reshape(-1).to(i1.device)[i1]); q2.append(QB.reshape(-1).to(i2.device)[i2])
    H1, W1 = feat11.shape[:2]; H2, W2 = feat22.shape[:2]
    xy1, xy2, index = matching.merge_corres(torch.cat(idx1), torch.cat(idx2), (H1, W1), (H2, W2), ret_xy=True,
                                            ret_index=True)
    confs = (torch.cat(q1)[index] * torch.cat(q2)[index]).sqrt()
    return xy1.float(), xy2.float(), confs


def _pair_files(cache_path, n1, n2, desc_conf, subsample):
    i1, i2 = hash_md5(n1), hash_md5(n2)
    cdir = os.path.join(cache_path, f"corres_conf={desc_conf}_subsample={subsample}")
    return (os.path.join(cache_path, "forward", i1, i2 + ".pth"), os.path.join(cache_path, "forward", i2, i1 + ".pth"),
            os.path.join(cdir, f"{i1}-{i2}.pth"), os.path.join(cdir, f"{i2}-{i1}.pth"))


def _infer_pair(img1, img2, model, files, desc_conf, device, subsample):
    """One symmetric inference + the four reciprocal matchings of a pair -> the three cache entries (CPU tensors)."""
    res = model.symmetric_inference(img1, img2, device)
    X11, X21, X22, X12 = [r["pts3d"][0] for r in res]
    C11, C21, C22, C12 = [r["conf"][0] for r in res]
    descs = [r["desc"][0] for r in res]
    qonfs = [r[desc_conf][0] for r in res]
    cpu = lambda ts: tuple(t.detach().cpu() for t in ts)
    corres = extract_correspondences(descs, qonfs, subsample=subsample, device=device)
    conf_score = (C11.mean() * C12.mean() * C21.mean() * C22.mean()).sqrt().sqrt()
    score = (float(conf_score), float(corres[2].sum()), len(corres[2]))
    return {files[0]: cpu((X11, C11, X21, C21)), files[1]: cpu((X22, C22, X12, C12)), files[2]: (score, cpu(corres))}


def _exchange_pair_results(mine, cache_path, device):
    """Every rank ends up with every rank's new cache entries.  The payload (pointmaps, confidences, the
    variable-length correspondence lists) travels as ONE flat float32 vector per rank through
    dist.all_gather_varlen; a small object all-gather carries the layout (relative paths, shapes, scores)."""
    import torch.distributed as tdist
    from . import dist as sdist
    layout, chunks = [], []
    for path, obj in mine.items():
        rel = os.path.relpath(path, cache_path)
        if isinstance(obj[0], tuple):      # correspondence entry: ((score, sum, count), (xy1, xy2, confs))
            score, tensors = obj
        else:
            score, tensors = None, obj
        # the payload travels as float32 (exact for the float16 / integer tensors a cache entry can hold: |values| < 2^24);
        # the layout carries each tensor's dtype so that the receiver stores what the sender stored
        layout.append((rel, score, [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors]))
        chunks.extend(t.reshape(-1).float() for t in tensors)
    flat = torch.cat(chunks) if chunks else torch.zeros(0)
    backend_dev = device if tdist.get_backend() == "nccl" else "cpu"
    parts = sdist.all_gather_varlen(flat.to(backend_dev))
    layouts = [None] * tdist.get_world_size()
    tdist.all_gather_object(layouts, layout)
    for r, (lay, buf) in enumerate(zip(layouts, parts)):
        if r == tdist.get_rank():
            continue
        buf = buf.cpu(); off = 0
        for rel, score, shapes in lay:
            tensors = []
            for shp, dt in shapes:
                n = 1
                for d in shp:
                    n *= d
                tensors.append(buf[off:off + n].reshape(shp).to(getattr(torch, dt)).clone()); off += n
            path = os.path.join(cache_path, rel)
            if not os.path.isfile(path):
                torch.save(tuple(tensors) if score is None else (score, tuple(tensors)), _mkdir_for(path))


def forward_mast3r(pairs, model, cache_path, desc_conf="desc_conf", device="cuda:0", subsample=8, shard=True,
                   **matching_kw):
    """pairs: iterable of (img1, img2) dicts with 'instance' (Mast3r's pair list).  Returns (res_paths, cache_path)
    with res_paths[(instance1, instance2)] = ((path1, path2), path_corres) -- the `tmp_pairs` of
    prepare_canonical_data.  Pairs already in the cache (in either order) are not inferred again.

    Under torch.distributed (one process per GPU) the pairs that still need inference are dealt round-robin to the
    ranks (the pairs are independent: no collective on the data path); afterwards the new cache entries are
    all-gathered (variable length: the correspondence lists differ per pair) so that every rank's cache -- shared
    directory or rank-private -- is complete and every rank returns the same res_paths.  shard=False: every rank
    infers every pair (the single-process behaviour)."""
    from . import dist as sdist
    rank, world = sdist.rank_world()
    if not shard:
        rank, world = 0, 1
    pairs = list(pairs)
    res_paths, todo = {}, []
    for img1, img2 in pairs:
        n1, n2 = img1["instance"], img2["instance"]
        if (n2, n1) in res_paths or (n1, n2) in res_paths:
            continue   # the symmetrized list holds both orders; one symmetric inference serves both
        files = _pair_files(cache_path, n1, n2, desc_conf, subsample)
        path1, path2, path_corres, path_corres2 = files
        if os.path.isfile(path_corres2) and not os.path.isfile(path_corres):
            score, (xy1, xy2, confs) = torch.load(path_corres2)
            torch.save((score, (xy2, xy1, confs)), _mkdir_for(path_corres))
        res_paths[n1, n2] = (path1, path2), path_corres
        if not all(os.path.isfile(p) for p in (path1, path2, path_corres)):
            todo.append((img1, img2, files))
    mine = {}
    if world > 1:
        # Every rank must deal from the SAME list.  Rank-private caches can be in different states (and a shared one
        # can change while the ranks take their inventory), so the ranks exchange the positions of the pairs they miss
        # and deal the union: a pair missing anywhere is inferred once, by one rank, and its entries reach every cache
        # through the exchange below (a rank that already holds a pair keeps its own files).
        import torch.distributed as tdist
        order = {}
        for img1, img2 in pairs:
            order.setdefault((img1["instance"], img2["instance"]), len(order))
        by_key = {(a["instance"], b["instance"]): (a, b) for a, b in pairs}
        missing = sorted(order[(a["instance"], b["instance"])] for a, b, _ in todo)
        everyone = [None] * tdist.get_world_size()
        tdist.all_gather_object(everyone, missing)
        union = sorted(set(i for lst in everyone for i in lst))
        keys = {v: k for k, v in order.items()}
        todo = []
        for i in union:
            img1, img2 = by_key[keys[i]]
            todo.append((img1, img2, _pair_files(cache_path, img1["instance"], img2["instance"], desc_conf, subsample)))
    if model is not None:
        for k in sdist.shard_pairs(len(todo), rank, world):
            img1, img2, files = todo[k]
            entries = _infer_pair(img1, img2, model, files, desc_conf, device, subsample)
            for path, obj in entries.items():
                torch.save(obj, _mkdir_for(path))
            mine.update(entries)
        if world > 1:
            _exchange_pair_results(mine, cache_path, device)
    # pairs that could not be completed (no model, nothing cached) are not reported
    res_paths = {k: v for k, v in res_paths.items() if all(os.path.isfile(p) for p in (*v[0], v[1]))}
    return res_paths, cache_path
