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
from starst3r_amd import synth

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
