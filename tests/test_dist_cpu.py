"""Multi-process (gloo, world_size 2, CPU) test of the view-sharded data-parallel step:
gradients after the all-reduce equal the single-process gradients over the union of views, including the
per-view regulariser accounting (starster/gs.py:150-152), and replicas stay identical after the update.
The per-rank "local step" is the CPU oracle here (no GPU in this container); on the GPU box the very same
starst3r_amd.dist helpers wrap the HIP train step (bench.py / gs.run_3dgs_optim)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import gs_oracle as go
from starst3r_amd import dist as sdist
from st3r_synth import synth

N, V, W, H = 120, 4, 48, 32


def scene():
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=12, scale_lo=0.03, scale_hi=0.15)
    gt_g = synth.perturb_for_gt(g, sigma=0.02)
    gt, _, _ = go.rasterization(gt_g["means"], gt_g["quats"], gt_g["scales"], gt_g["opacities"], gt_g["shN"], w2c, Ks, W, H)
    return g, w2c, Ks, np.clip(gt, 0, 1)


def oracle_local_step(g, w2c, Ks, gt, views, ssim_fac=0.2, opac_fac=0.01, scale_fac=0.01):
    """grads [23N] (block layout) and loss of the views of one rank, regularisers added once per local view."""
    grads = np.zeros(23 * N); loss = 0.0
    vw, vk = w2c[views], Ks[views]
    rgb, alpha, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], vw, vk, W, H)
    v_rgb = np.zeros_like(rgb)
    for c, v in enumerate(views):
        l1, ss, vr = go.l1_ssim(rgb[c], gt[v], 1 - ssim_fac, ssim_fac)
        loss += (1 - ssim_fac) * l1 + ssim_fac * (1 - ss); v_rgb[c] = vr
    G = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], vw, vk, W, H, meta,
                                  alpha, v_rgb, None)
    sg = 1 / (1 + np.exp(-g["opacities"].astype(np.float64))); ex = np.exp(g["scales"].astype(np.float64))
    k = len(views)
    loss += k * (opac_fac * sg.mean() + scale_fac * ex.mean())
    G["opacities"] = G["opacities"] + k * opac_fac * sg * (1 - sg) / N
    G["scales"] = G["scales"] + k * scale_fac * ex / (3 * N)
    off = 0
    for key, w in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1), ("sh", 12)):
        grads[off * N:(off + w) * N] = np.asarray(G[key]).reshape(-1); off += w
    return grads, loss


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    g, w2c, Ks, gt = scene()
    assert sdist.rank_world() == (rank, world)
    views = sdist.shard_views(V, rank, world)
    grads = torch.zeros(23 * N, dtype=torch.float64); loss = torch.zeros(1, dtype=torch.float64)

    def local(gr, ls):
        a, b = oracle_local_step(g, w2c, Ks, gt, views)
        gr.copy_(torch.from_numpy(a)); ls[0] = b
    sdist.sharded_step(local, grads, loss)
    # identical replica update (plain SGD stands in for the fused Adam): parameters must match bit for bit
    p = torch.from_numpy(g["means"].astype(np.float64).reshape(-1)) - 1e-3 * grads[:3 * N]
    gathered = [torch.zeros_like(p) for _ in range(world)]
    torch.distributed.all_gather(gathered, p)
    assert all(torch.equal(gathered[0], x) for x in gathered)
    if rank == 0:
        np.savez(out, grads=grads.numpy(), loss=loss.numpy(), views=np.array(views))
    torch.distributed.destroy_process_group()


def test_view_sharded_gradients_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    g, w2c, Ks, gt = scene()
    ref_grads, ref_loss = oracle_local_step(g, w2c, Ks, gt, list(range(V)))
    assert z["views"].tolist() == [0, 2]
    np.testing.assert_allclose(z["loss"][0], ref_loss, rtol=1e-12)
    np.testing.assert_allclose(z["grads"], ref_grads, rtol=1e-9, atol=1e-12)


def test_shard_views_partitions_and_rejects_idle_ranks():
    parts = [sdist.shard_views(8, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(8)) and all(len(p) == 2 for p in parts)
    with pytest.raises(ValueError):
        sdist.shard_views(2, 3, 4)
    assert sdist.rank_world() == (0, 1)


# ---- Gaussian-sharded mode: the two all-to-all exchanges (layout logic) on gloo, world size 2 ----

def _label(rank_of_gaussian, view, g_local, K=3):
    return torch.tensor([float(rank_of_gaussian), float(view), float(g_local)])[:K]


def _a2a_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    V, n, K = 4, 5, 3                                   # 4 views -> C = 2 per rank; 5 Gaussians per rank
    C = V // world
    assert sdist.shard_gaussians(n * world, rank, world) == (rank * n, (rank + 1) * n)
    assert sdist.shard_views_contiguous(V, rank, world) == list(range(rank * C, (rank + 1) * C))
    local = torch.stack([torch.stack([_label(rank, v, i) for i in range(n)]) for v in range(V)])     # [V, n, K]
    mine = sdist.records_to_view_owners(local, world)                                                  # [C, world*n, K]
    ok = mine.shape == (C, world * n, K)
    for c in range(C):
        for g in range(world * n):
            ok &= bool(torch.equal(mine[c, g], _label(g // n, rank * C + c, g % n)))
    back = sdist.records_to_gaussian_owners(mine * 2.0, world)                                         # [V, n, K]
    ok &= bool(torch.equal(back, local * 2.0))
    with open(f"{out}.{rank}", "w") as f:
        f.write("ok" if ok else "bad")
    torch.distributed.destroy_process_group()


def test_sharded_exchanges_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "a2a")
    mp.spawn(_a2a_worker, args=(2, port, out), nprocs=2, join=True)
    assert [open(f"{out}.{r}").read() for r in range(2)] == ["ok", "ok"]


def _a2a_uneven_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    V, N, K = 4, 11, 3                                  # 11 Gaussians over 2 ranks: shards of 5 and 6
    C = V // world
    counts = sdist.shard_counts(N, world)
    lo, hi = sdist.shard_gaussians(N, rank, world)
    ok = counts == [5, 6] and hi - lo == counts[rank] and (lo, hi) == ((0, 5), (5, 11))[rank]
    n = hi - lo
    owner = lambda g: 0 if g < counts[0] else 1
    local = torch.stack([torch.stack([_label(rank, v, i) for i in range(n)]) for v in range(V)])     # [V, n, K]
    mine = sdist.records_to_view_owners(local, world, counts=counts)                                   # [C, N, K]
    ok &= mine.shape == (C, N, K)
    for c in range(C):
        for g in range(N):
            r = owner(g)
            ok &= bool(torch.equal(mine[c, g], _label(r, rank * C + c, g - (0, counts[0])[r])))
    back = sdist.records_to_gaussian_owners(mine * 2.0, world, counts=counts, rank=rank)               # [V, n, K]
    ok &= bool(torch.equal(back, local * 2.0))
    with open(f"{out}.{rank}", "w") as f:
        f.write("ok" if ok else "bad")
    torch.distributed.destroy_process_group()


def test_sharded_exchanges_uneven_shards_world2(tmp_path):
    """N not divisible by the ranks (the MCMC growth steps leave such N): all_to_all_single with split sizes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "a2au")
    mp.spawn(_a2a_uneven_worker, args=(2, port, out), nprocs=2, join=True)
    assert [open(f"{out}.{r}").read() for r in range(2)] == ["ok", "ok"]


def _gather_rows_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from starst3r_amd import gs
    ok = True
    for N in (10, 11):                                   # even and uneven shards
        counts = sdist.shard_counts(N, world)
        lo, hi = sdist.shard_gaussians(N, rank, world)
        for width in (1, 3):
            truth = torch.arange(N * width, dtype=torch.float32).reshape(N, width) * 0.5
            full = torch.full((N, width), -1.0)
            full[lo:hi] = truth[lo:hi]                   # every rank knows only its own rows ...
            gs._gather_rows(full, full[lo:hi], counts, width)   # ... and passes a VIEW of the output as its input
            ok &= bool(torch.equal(full, truth))
    with open(f"{out}.{rank}", "w") as f:
        f.write("ok" if ok else "bad")
    torch.distributed.destroy_process_group()


def test_sharded_rows_are_gathered_even_and_uneven_world2(tmp_path):
    """gs._gather_rows (the re-assembly of the Gaussian shards before an MCMC refinement step and at the end of
    run_3dgs_optim on the sharded layout): equal shards through all_gather_into_tensor, uneven ones through the
    variable-length all-gather; the local rows are a view of the output tensor."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "gr")
    mp.spawn(_gather_rows_worker, args=(2, port, out), nprocs=2, join=True)
    assert [open(f"{out}.{r}").read() for r in range(2)] == ["ok", "ok"]


# ---------------------------------------------------------------------------------------------------------------
# Path A under torch.distributed: the pairs of the complete graph are dealt round-robin to the ranks and the new
# cache entries all-gathered with variable lengths (starst3r_amd.forward.forward_mast3r).
# ---------------------------------------------------------------------------------------------------------------
def _cpu_extract_correspondences(feats, qonfs, subsample=8, device="cpu"):
    """stand-in for the MFMA matching on this GPU-less box: the numpy oracle, same merge and confidence rule"""
    from oracle import nn_oracle
    from starst3r_amd import matching
    f = [x.cpu().numpy() for x in feats]; q = [x.cpu().numpy() for x in qonfs]
    idx1, idx2, q1, q2 = [], [], [], []
    for A, B, QA, QB in ((f[0], f[1], q[0], q[1]), (f[3], f[2], q[3], q[2])):
        a12, b12 = nn_oracle.fast_reciprocal_NNs(A, B, subsample)
        b21, a21 = nn_oracle.fast_reciprocal_NNs(B, A, subsample)
        i1 = np.concatenate([a12, a21]).astype(np.int64); i2 = np.concatenate([b12, b21]).astype(np.int64)
        idx1.append(i1); idx2.append(i2); q1.append(QA.reshape(-1)[i1]); q2.append(QB.reshape(-1)[i2])
    H1, W1 = f[0].shape[:2]; H2, W2 = f[2].shape[:2]
    xy1, xy2, index = matching.merge_corres(torch.tensor(np.concatenate(idx1)), torch.tensor(np.concatenate(idx2)),
                                            (H1, W1), (H2, W2), ret_xy=True, ret_index=True)
    confs = torch.tensor(np.sqrt(np.concatenate(q1) * np.concatenate(q2)))[index]
    return xy1.float(), xy2.float(), confs.float()


def _pairs_of(n):
    imgs = [dict(instance=f"{i}.png", idx=i) for i in range(n)]
    return [(imgs[i], imgs[j]) for i in range(n) for j in range(i + 1, n)]


def _forward_worker(rank, world, port, base, uneven=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from starst3r_amd import forward
    from st3r_synth.synth_model import SyntheticNetwork
    forward.extract_correspondences = _cpu_extract_correspondences
    net = SyntheticNetwork(n_views=4, width=64, height=48, seed=1)
    cache = os.path.join(base, f"rank{rank}")            # rank-private caches: the exchange has to fill them
    if uneven and rank == 1:
        # the caches start in DIFFERENT states: rank 1 already holds the pairs among views 0..2 (the ranks
        # used to deal from their own lists of missing pairs; now they deal the union of what is missing anywhere)
        sub = [p for p in _pairs_of(4) if p[0]["idx"] < 3 and p[1]["idx"] < 3]
        forward.forward_mast3r(sub, net, cache, device="cpu", subsample=8, shard=False)
        net.calls = 0
    res, _ = forward.forward_mast3r(_pairs_of(4), net, cache, device="cpu", subsample=8)
    assert net.calls == 3, net.calls                     # 6 pairs over 2 ranks
    assert len(res) == 6
    torch.save({k: (torch.load(v[0][0]), torch.load(v[0][1]), torch.load(v[1])) for k, v in res.items()},
               os.path.join(base, f"out{rank}.pth"))
    # a second call finds everything cached: no inference, no payload
    res2, _ = forward.forward_mast3r(_pairs_of(4), net, cache, device="cpu", subsample=8)
    assert net.calls == 3 and list(res2) == list(res)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("uneven", [False, True])
def test_pair_sharding_fills_every_ranks_cache(tmp_path, uneven):
    from starst3r_amd import forward
    from st3r_synth.synth_model import SyntheticNetwork
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_forward_worker, args=(2, port, str(tmp_path), uneven), nprocs=2, join=True)
    # single-process reference with the same stand-in matcher
    keep = forward.extract_correspondences
    forward.extract_correspondences = _cpu_extract_correspondences
    try:
        net = SyntheticNetwork(n_views=4, width=64, height=48, seed=1)
        res, _ = forward.forward_mast3r(_pairs_of(4), net, str(tmp_path / "single"), device="cpu", subsample=8)
    finally:
        forward.extract_correspondences = keep
    assert net.calls == 6
    outs = [torch.load(tmp_path / f"out{r}.pth") for r in range(2)]

    def same(a, b):
        if torch.is_tensor(a):
            return torch.equal(a, b)
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b
    for k, v in res.items():
        want = (torch.load(v[0][0]), torch.load(v[0][1]), torch.load(v[1]))
        for r in range(2):
            assert same(outs[r][k], want), (k, r)
    assert same(sdist.all_gather_varlen(torch.arange(5.0)), [torch.arange(5.0)])   # single process: identity


# ---------------------------------------------------------------------------------------------------------------
# The rank bookkeeping of gs.run_3dgs_optim itself (view shard, per-view regularisers, gradient / loss all-reduce,
# step counter, identical replica updates) with only the C calls replaced by the CPU oracle.
# ---------------------------------------------------------------------------------------------------------------
class _CpuScene:
    def __init__(self, g, w2c, Ks, gt):
        self.device = "cpu"
        self.imgs = [gt[v] for v in range(gt.shape[0])]
        self._w2c = torch.tensor(w2c); self.intrinsics = torch.tensor(Ks)
        self.dense_pts = [torch.tensor(g["means"])]; self.dense_cols = [torch.zeros(N, 3)]

    dense_pts_flat = property(lambda self: self.dense_pts[0])
    dense_cols_flat = property(lambda self: self.dense_cols[0])
    w2c = property(lambda self: self._w2c)


def _mock_ops(monkey_ops, g_keys=("means", "quats", "scales", "opacities", "shN")):
    """replace the two C calls of the non-fused branch of run_3dgs_optim by the oracle"""
    calls = dict(fwd_bwd=0, adam=0)

    def train_fwd_bwd(ctx, P, w2c, Ks, campos, gt, W_, H_, ssim_fac, opac_fac, scale_fac, grads, loss_out,
                      want_stats=True):
        calls["fwd_bwd"] += 1
        g = {k: P[k].numpy() for k in g_keys}
        C = w2c.shape[0]
        gr, ls = oracle_local_step(g, w2c.numpy(), Ks.numpy(), gt.numpy(), list(range(C)), ssim_fac, opac_fac, scale_fac)
        grads.copy_(torch.from_numpy(gr).to(grads.dtype)); loss_out[0] = ls
        return dict(n_visible=0, n_isects=0, n_isects_ref=0, arena_bytes=0)

    def adam_step(ctx, P, grads, m, v, lr, b1, b2, eps, step):
        calls["adam"] += 1
        off = 0
        for key, w in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1)):
            n = w * N
            p = P[key].numpy().reshape(-1); mm = m[off:off + n].numpy(); vv = v[off:off + n].numpy()
            go.adam(p, grads[off:off + n].numpy(), mm, vv, lr, b1, b2, eps, step); off += n
        p = np.ascontiguousarray(P["shN"].numpy()[:, :4]).reshape(-1); n = 12 * N
        mm = m[off:off + n].numpy(); vv = v[off:off + n].numpy()
        go.adam(p, grads[off:off + n].numpy(), mm, vv, lr, b1, b2, eps, step)
        P["shN"][:, :4] = torch.from_numpy(p.reshape(N, 4, 3))
    def train_step(ctx, P, w2c, Ks, campos, gt, W_, H_, ssim_fac, opac_fac, scale_fac, grads, m, v, lr, b1, b2, eps,
                   step, loss_out, want_stats=True):   # the fused single-process call: the same two pieces, no exchange
        st = train_fwd_bwd(ctx, P, w2c, Ks, campos, gt, W_, H_, ssim_fac, opac_fac, scale_fac, grads, loss_out)
        adam_step(ctx, P, grads, m, v, lr, b1, b2, eps, step)
        return st
    monkey_ops.train_fwd_bwd = train_fwd_bwd
    monkey_ops.adam_step = adam_step
    monkey_ops.train_step = train_step
    monkey_ops.get_context = lambda device: type("Ctx", (), {"native_comm": False, "device": "cpu"})()
    monkey_ops.settle = lambda ctx: None     # the stand-in steps are synchronous: nothing is ever in flight
    # (round 6: the loop registers SSIM's ground-truth moments with the ctx once per call -- C calls like the others; the
    # oracle stand-in of the step convolves the ground truth itself)
    monkey_ops.gt_moments = lambda ctx, gt: None
    monkey_ops.set_gt_moments = lambda ctx, gt, mom: None
    return calls


def _run_optim(world_tag):
    from starst3r_amd import gs, ops
    calls = _mock_ops(ops)
    g, w2c, Ks, gt = scene()
    sc = _CpuScene(g, w2c, Ks, gt.astype(np.float32))
    gs.init_3dgs(sc)
    with torch.no_grad():
        for k in ("means", "quats", "scales", "opacities", "shN"):
            sc.gaussians[k].data.copy_(torch.tensor(g[k]))
    losses = gs.run_3dgs_optim(sc, 3)
    return sc, losses, calls


def _optim_worker(rank, world, port, base):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    sc, losses, calls = _run_optim(f"w{world}")
    assert calls == dict(fwd_bwd=3, adam=3) and sc._gs_optim.step == 3
    torch.save(dict(losses=losses, **{k: v.data.clone() for k, v in sc.gaussians.items()}),
               os.path.join(base, f"optim{rank}.pth"))
    torch.distributed.destroy_process_group()


def test_run_3dgs_optim_rank_bookkeeping_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_optim_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"optim{r}.pth") for r in range(2))
    sc, losses, calls = _run_optim("single")
    assert calls == dict(fwd_bwd=3, adam=3)
    for k in ("means", "quats", "scales", "opacities", "shN"):
        assert torch.equal(r0[k], r1[k]), k                                   # replicas identical, bit for bit
        np.testing.assert_allclose(r0[k].numpy(), sc.gaussians[k].data.numpy(), rtol=0, atol=2e-6, err_msg=k)
    assert r0["losses"] == r1["losses"] and len(r0["losses"]) == 3           # every rank returns the summed loss
    np.testing.assert_allclose(r0["losses"], losses, rtol=1e-6)
