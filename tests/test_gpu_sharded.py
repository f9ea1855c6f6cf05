"""Gaussian-sharded multi-GPU mode on ONE GPU: `world` virtual ranks run as threads (each with its own st3r ctx),
the all-to-all is emulated with a barrier, and the result is compared with the single-process fused step --
gradients after the first iteration and parameters after several.  The real exchange (torch.distributed
all_to_all_single) is covered on gloo by tests/test_dist_cpu.py::test_sharded_exchanges_world2."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from st3r_synth import synth

DEV = torch.device("cuda:0")


class A2A:
    """all_to_all_single among threads: chunk r of rank s's send buffer lands in chunk s of rank r's recv buffer."""

    def __init__(self, world):
        self.world, self.box, self.bar = world, [None] * world, threading.Barrier(world)

    def rank_fn(self, rank):
        def a2a(recv, send, recv_splits=None, send_splits=None):
            k = send.numel() // self.world
            self.box[rank] = (send.clone(), send_splits if send_splits is not None else [k] * self.world)
            torch.cuda.synchronize()
            self.bar.wait()
            off = 0
            for s in range(self.world):
                buf, splits = self.box[s]
                src = sum(splits[:rank])
                cnt = splits[rank]
                assert recv_splits is None or recv_splits[s] == cnt
                recv[off:off + cnt].copy_(buf[src:src + cnt])
                off += cnt
            torch.cuda.synchronize()
            self.bar.wait()
        return a2a


@pytest.mark.parametrize("world,V,N", [(2, 4, 6000), (4, 4, 6000), (2, 2, 6000), (2, 4, 6001), (4, 8, 6002)])
def test_sharded_equals_replicated(world, V, N):
    """N not divisible by the ranks: balanced shards whose sizes differ by one (what the MCMC growth leaves behind)."""
    from starst3r_amd import dist as sdist, ops
    W, H, steps = 160, 96, 6
    g, w2c_np, Ks_np = synth.make_scene(N, V, W, H, seed=9, scale_lo=0.01, scale_hi=0.05)
    P0 = {k: torch.from_numpy(g[k]).to(DEV) for k in ("means", "quats", "scales", "opacities", "shN")}
    w2c = torch.from_numpy(w2c_np).to(DEV); Ks = torch.from_numpy(Ks_np).to(DEV)
    ctx0 = ops.get_context(DEV)
    campos = ops.camera_positions(w2c)
    gt_g = synth.perturb_for_gt(g, sigma=0.01)
    Q = {k: torch.from_numpy(gt_g[k]).to(DEV) for k in P0}
    gt, _, _ = ops.render(ctx0, Q, w2c, Ks, campos, W, H)
    gt = gt.clamp(0, 1).contiguous()
    # reference: single process, all views, fused step
    A = {k: v.clone() for k, v in P0.items()}
    grads = torch.empty(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    ref_losses = torch.zeros(steps, device=DEV); first_grads = None
    for t in range(steps):
        ops.train_step(ctx0, A, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, t + 1,
                       ref_losses[t:t + 1])
        if t == 0:
            first_grads = grads.clone()
    torch.cuda.synchronize()
    # sharded: `world` virtual ranks
    bus = A2A(world)
    counts = sdist.shard_counts(N, world)
    shards, losses, g1, errors = [None] * world, [None] * world, [None] * world, []

    def run(rank):
        try:
            ctx = ops.Context(DEV)
            lo, hi = sdist.shard_gaussians(N, rank, world)
            views = sdist.shard_views_contiguous(V, rank, world)
            P = {k: P0[k][lo:hi].clone() for k in P0}
            tr = sdist.ShardedTrainer(ctx, P, N, w2c, Ks, gt[views].contiguous(), W, H, rank, world, a2a=bus.rank_fn(rank))
            L = torch.zeros(steps, device=DEV)
            for t in range(steps):
                tr.step(L[t:t + 1])
                if t == 0:
                    g1[rank] = tr.grads.clone()
            torch.cuda.synchronize()
            shards[rank], losses[rank] = P, L
        except Exception as e:  # noqa: BLE001 -- surfaced below; a dead thread would deadlock the barrier otherwise
            errors.append(e); bus.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    # gradients of the first iteration: shard blocks vs the corresponding rows of the full gradient
    off = 0
    for name, wdt in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1), ("sh", 12)):
        full = first_grads[off * N:(off + wdt) * N].reshape(N, wdt)
        got = torch.cat([g1[r][off * counts[r]:(off + wdt) * counts[r]].reshape(counts[r], wdt) for r in range(world)])
        scale = float(full.abs().max())
        assert float((got - full).abs().max()) <= 2e-4 * scale + 1e-12, name
        off += wdt
    # the loss is a sum over ranks
    total = sum(losses)
    np.testing.assert_allclose(total.cpu().numpy(), ref_losses.cpu().numpy(), rtol=2e-5)
    # parameters after `steps` Adam updates (each update moves a parameter by <= lr; sign flips of ~zero gradients
    # are the only way to differ by more than rounding)
    for k in P0:
        got = torch.cat([shards[r][k] for r in range(world)])
        d = (got - A[k]).abs()
        assert float(d.max()) <= 2.5e-3 and float((d > 1e-5).float().mean()) < 0.01, k


def test_sharded_step_through_rccl_single_rank():
    """The exchange as the multi-GPU run issues it -- torch.distributed.all_to_all_single on the nccl (= RCCL) backend
    -- with the one rank a single-GPU box can host: same numbers as the copy-based exchange."""
    import os
    import torch.distributed as dist
    from starst3r_amd import dist as sdist, ops
    N, V, W, H = 4000, 2, 128, 96
    g, w2c_np, Ks_np = synth.make_scene(N, V, W, H, seed=4, scale_lo=0.01, scale_hi=0.05)
    P0 = {k: torch.from_numpy(g[k]).to(DEV) for k in ("means", "quats", "scales", "opacities", "shN")}
    w2c = torch.from_numpy(w2c_np).to(DEV); Ks = torch.from_numpy(Ks_np).to(DEV)
    ctx = ops.get_context(DEV)
    gt = torch.rand((V, H, W, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))

    def run(a2a):
        P = {k: v.clone() for k, v in P0.items()}
        tr = sdist.ShardedTrainer(ctx, P, N, w2c, Ks, gt, W, H, 0, 1, a2a=a2a)
        L = torch.zeros(3, device=DEV)
        for t in range(3):
            tr.step(L[t:t + 1])
        torch.cuda.synchronize()
        return P, L

    ref_P, ref_L = run(lambda recv, send: recv.copy_(send))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        got_P, got_L = run(sdist._all_to_all)
    finally:
        dist.destroy_process_group()
    np.testing.assert_allclose(got_L.cpu().numpy(), ref_L.cpu().numpy(), rtol=1e-6)
    for k in ref_P:
        assert float((got_P[k] - ref_P[k]).abs().max()) <= 2.5e-3 and float(((got_P[k] - ref_P[k]).abs() > 1e-5).float().mean()) < 0.01


def test_scene_run_3dgs_optim_sharded_layout_equals_replicated(monkeypatch):
    """Scene.run_3dgs_optim on the Gaussian-sharded layout (forced with one rank) leaves the same parameters, optimiser
    state and losses as the default single-process path, starting from the same scene state."""
    import starst3r_amd as st
    from st3r_synth.synth_model import SyntheticPairwiseModel
    model = SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2)
    sc = st.Scene(device="cuda:0")
    sc.add_images(model, [torch.zeros(3, 96, 128) for _ in range(2)])
    sc.init_3dgs()
    keys = ("means", "quats", "scales", "opacities", "shN")
    start = {k: sc.gaussians[k].data.clone() for k in keys}
    la = sc.run_3dgs_optim(6)
    A = {k: sc.gaussians[k].data.clone() for k in keys}
    mA = sc._gs_optim.m.clone()
    # rewind to the same starting state
    for k in keys:
        sc.gaussians[k].data.copy_(start[k])
    sc._gs_optim.m.zero_(); sc._gs_optim.v.zero_(); sc._gs_optim.step = 0
    monkeypatch.setenv("ST3R_MULTI_GPU", "gaussian-sharded")
    lb = sc.run_3dgs_optim(4); lb += sc.run_3dgs_optim(2)      # two calls: the optimiser state carries over
    monkeypatch.delenv("ST3R_MULTI_GPU")
    np.testing.assert_allclose(lb, la, rtol=2e-5)
    assert sc._gs_optim.step == 6
    for k in keys:
        d = (A[k] - sc.gaussians[k].data).abs()
        assert float(d.max()) <= 2.5e-3 and float((d > 1e-5).float().mean()) < 0.01, k
    dm = (mA - sc._gs_optim.m).abs()
    assert float(dm.max()) <= 1e-3 * float(mA.abs().max()) + 1e-9


def test_noise_of_a_shard_is_the_noise_of_its_rows():
    """st3r_mcmc_noise_rows keys the draws by the global row: perturbing rows [lo, hi) with row_offset = lo equals the
    same rows of a whole-set call, bit for bit."""
    from starst3r_amd import ops
    N, lo, hi = 5000, 1234, 4321
    g, _, _ = synth.make_scene(N, 1, 64, 64, seed=3)
    P = {k: torch.from_numpy(g[k]).to(DEV) for k in ("means", "quats", "scales", "opacities")}
    P["scales"] = torch.log(P["scales"]); P["opacities"] = torch.logit(P["opacities"].clamp(1e-4, 0.5))
    ctx = ops.get_context(DEV)
    whole = {k: v.clone() for k, v in P.items()}
    ops.mcmc_noise(ctx, whole, 5e2, 17, 9)
    part = {k: v[lo:hi].clone() for k, v in P.items()}
    ops.mcmc_noise(ctx, part, 5e2, 17, 9, row_offset=lo)
    assert not torch.equal(whole["means"], P["means"])
    assert torch.equal(part["means"], whole["means"][lo:hi])


def test_scene_sharded_layout_with_pruning_equals_replicated(monkeypatch):
    """enable_pruning on the Gaussian-sharded layout (forced with one rank; refinement window pulled to the front):
    gather -> relocate / grow on the full set -> shard again, noise per shard -- the same Gaussian count, losses and
    parameters as the replicated loop."""
    import starst3r_amd as st
    from st3r_synth.synth_model import SyntheticPairwiseModel
    keys = ("means", "quats", "scales", "opacities", "shN")

    def run(sharded):
        model = SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2)
        sc = st.Scene(device="cuda:0")
        sc.add_images(model, [torch.zeros(3, 96, 128) for _ in range(2)])
        sc.init_3dgs()
        sc.strategy.refine_start_iter, sc.strategy.refine_every = 1, 3
        sc.strategy.cap_max = int(sc.gaussians["means"].shape[0] * 1.08)
        with torch.no_grad():   # some dead Gaussians for the relocation
            sc.gaussians["opacities"].data[::7] = -8.0
        if sharded:
            monkeypatch.setenv("ST3R_MULTI_GPU", "gaussian-sharded")
        try:
            losses = sc.run_3dgs_optim(8, enable_pruning=True)
        finally:
            monkeypatch.delenv("ST3R_MULTI_GPU", raising=False)
        return sc, losses
    a, la = run(False)
    b, lb = run(True)
    assert a.gaussians["means"].shape == b.gaussians["means"].shape
    assert a.strategy_state["calls"] == b.strategy_state["calls"] == 8
    assert a.strategy_state["n_added"] == b.strategy_state["n_added"]
    np.testing.assert_allclose(lb, la, rtol=1e-4)
    for k in keys:
        d = (a.gaussians[k].data - b.gaussians[k].data).abs()
        assert float((d > 1e-4).float().mean()) < 0.01, k
    assert a._gs_optim.step == b._gs_optim.step == 8
