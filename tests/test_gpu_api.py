"""End-to-end drop-in API test on the GPU (BASELINE configs[0]/[1] in miniature): Scene.add_images with a
synthetic pairwise model -> init_3dgs -> run_3dgs_optim -> render_3dgs, plus autograd through render_3dgs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as go


def test_scene_pipeline_end_to_end():
    import starst3r_amd as st
    from st3r_synth.synth_model import SyntheticPairwiseModel
    model = SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2)
    scene = st.Scene(device="cuda:0")
    raw = [torch.zeros(3, 96, 128) for _ in range(2)]
    scene.add_images(model, raw)                      # reconstruct: align on HIP (paths A/B entry point)
    assert scene.c2w.shape == (2, 4, 4) and scene.intrinsics.shape == (2, 3, 3) and len(scene.imgs) == 2
    assert set(scene.optim_params) >= {"pps", "log_focals", "quats", "trans", "log_sizes", "core_depth"}
    scene.add_images(model, [torch.zeros(3, 96, 128)])  # incremental: warm start from optim_params (B-9)
    assert scene.c2w.shape == (3, 4, 4) and len(scene.imgs) == 3
    scene.init_3dgs()
    g = scene.gaussians
    N = g["means"].shape[0]
    assert set(g) == {"means", "scales", "quats", "opacities", "sh0", "shN"} and g["shN"].shape == (N, 24, 3)
    assert float(g["scales"][0, 0]) == pytest.approx(3e-3) and float(g["quats"][0, 0]) == 1.0
    cols = scene.dense_cols_flat.to("cuda:0")
    assert torch.allclose(g["shN"][:, 7].data, 1 - cols)           # every SH row = 1 - colour (B-4)
    sh_tail = g["shN"].data[:, 4:].clone(); sh0 = g["sh0"].data.clone()
    losses = scene.run_3dgs_optim(40)
    assert len(losses) == 40 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert torch.equal(g["shN"].data[:, 4:], sh_tail) and torch.equal(g["sh0"].data, sh0)   # never updated (B-2/B-3)
    losses2 = scene.run_3dgs_optim(5, enable_pruning=True)           # noise injection path of MCMCStrategy
    assert len(losses2) == 5 and scene._gs_optim.step == 45          # Adam step counters persist (B-6)
    img, alpha, info = scene.render_3dgs_original(128, 96)
    assert img.shape == (3, 96, 128, 3) and alpha.shape == (3, 96, 128, 1)
    for k in ("camera_ids", "gaussian_ids", "radii", "means2d", "depths", "conics", "opacities", "tile_width",
              "tile_height", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "width", "height",
              "tile_size", "n_cameras"):
        assert k in info, k
    st_ = scene.optimizers["means"].state
    assert list(st_.values())[0]["step"] == 45 and scene.optimizers["sh0"].state == {}


def test_autograd_through_render_matches_oracle():
    import starst3r_amd as st
    from st3r_synth import synth
    N, V, W, H = 300, 2, 64, 48
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=9, scale_lo=0.02, scale_hi=0.1)
    scene = st.Scene(device="cuda:0")
    scene.gaussians = {k: torch.nn.Parameter(torch.tensor(v, device="cuda:0")) for k, v in g.items()}
    rgb, alpha, info = scene.render_3dgs(torch.tensor(w2c), torch.tensor(Ks), W, H)
    rng = np.random.default_rng(0)
    v_rgb = rng.standard_normal(rgb.shape).astype(np.float32)
    (rgb * torch.tensor(v_rgb, device="cuda:0")).sum().backward()
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
    ref = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H, meta,
                                    alpha_o, v_rgb, None)
    for k, rk in (("means", "means"), ("quats", "quats"), ("scales", "scales"), ("opacities", "opacities")):
        a = scene.gaussians[k].grad.cpu().numpy(); b = ref[rk]
        assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max() + 1e-9, k
    a = scene.gaussians["shN"].grad.cpu().numpy()
    assert np.abs(a[:, :4] - ref["sh"]).max() <= 1e-3 * np.abs(ref["sh"]).max() and np.abs(a[:, 4:]).max() == 0


def test_async_steps_equal_synchronous_steps_and_overflow_is_loud():
    """st3r_gs_train_fwd_bwd without stats_host keeps the record count on the device (no host synchronisation in
    steady state): gradients and losses of such steps are bit-identical to those of synchronous steps on the same
    inputs; a step that outgrows the capacity derived from the previous count makes the NEXT call fail with
    ST3R_ERR_CAPACITY, after which the context recovers on the synchronous path."""
    import numpy as np
    from starst3r_amd import _lib, ops
    from st3r_synth import synth
    ctx = ops.Context("cuda:0")     # a private context: the record-count hint is per context
    N, V, W, H = 20000, 3, 320, 240
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=5, scale_lo=0.004, scale_hi=0.03)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    P0 = {k: dev(v) for k, v in g.items()}
    rgb, _, _ = ops.render(ctx, P0, vm, K, campos, W, H)
    gt = torch.clamp(rgb + 0.05 * torch.randn_like(rgb), 0, 1).contiguous()

    def run(want_stats, steps=6):
        P = {k: v.clone() for k, v in P0.items()}
        grads = torch.empty(23 * N, device="cuda:0"); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        losses = torch.zeros(steps, device="cuda:0")
        snaps = []
        for it in range(steps):
            ops.train_step(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                           losses[it:it + 1], want_stats=want_stats or it == 0)   # the first step establishes the hint
            snaps.append(grads.clone())
        torch.cuda.synchronize()
        return P, snaps, losses
    Ps, gs_sync, ls = run(True)
    Pa, gs_async, la = run(False)
    for a, b in zip(gs_sync, gs_async):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert torch.equal(ls, la)
    for k in Ps:
        assert torch.equal(Ps[k], Pa[k]), k
    # overflow: the test hook halves the capacity of the asynchronous steps
    P = {k: v.clone() for k, v in P0.items()}
    grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)                     # sync: hint
    ops.set_debug(ctx, 8)
    ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss, want_stats=False)   # overflows quietly
    ops.set_debug(ctx, 0)
    with pytest.raises(_lib.St3rError) as e:
        ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss, want_stats=False)
    assert "more than" in str(e.value)
    ref = torch.empty_like(grads)
    ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, ref, loss, want_stats=False)     # recovered (sync)
    ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss, want_stats=False)   # async again
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int32), grads.view(torch.int32)) and torch.equal(ref.view(torch.int32), gs_sync[0].view(torch.int32))
    ctx.close()


@pytest.mark.gpu
def test_overflowing_async_step_writes_nothing_out_of_bounds_and_skips_its_update():
    """A FRESH context (its scratch was never sized for the full record count) takes one synchronous step, then an
    asynchronous step whose capacity the test hook halves: the step's slot indices (scan over the TRUE tile counts) exceed
    the slots the gradient buffer has -- the blend backward must not write them (out-of-bounds in
    k_blend_bwd / k_gather_vtile) --, its Adam update is skipped on the device, st3r_ctx_settle reports the overflow, and
    repeating the step gives exactly the parameters of an undisturbed run."""
    import numpy as np
    from starst3r_amd import _lib, ops
    from st3r_synth import synth
    N, V, W, H = 20000, 3, 320, 240
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=5, scale_lo=0.004, scale_hi=0.03)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    P0 = {k: dev(v) for k, v in g.items()}
    ref_ctx = ops.Context("cuda:0")
    rgb, _, _ = ops.render(ref_ctx, P0, vm, K, campos, W, H)
    gt = torch.clamp(rgb + 0.05 * torch.randn_like(rgb), 0, 1).contiguous()

    def steps(ctx, overflow_at):
        P = {k: v.clone() for k, v in P0.items()}
        grads = torch.empty(23 * N, device="cuda:0"); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        loss = torch.zeros(1, device="cuda:0")
        it = 1
        while it <= 3:
            before = {k: x.clone() for k, x in P.items()} if it == overflow_at else None
            if it == overflow_at:
                ops.set_debug(ctx, 8)
            # (no statistics asked for: the first step of a context still sizes its buffers exactly and leaves the hint the
            # asynchronous steps after it size theirs from)
            ops.train_step(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it, loss,
                           want_stats=False)
            ops.set_debug(ctx, 0)
            if it == overflow_at:
                overflow_at = -1
                with pytest.raises(_lib.St3rError) as e:
                    ops.settle(ctx)
                assert e.value.code == -3 and "NOT apply" in str(e.value)
                for k in P:   # the update of the overflowing step did not happen
                    assert torch.equal(P[k], before[k]), k
                continue      # repeat the same iteration
            it += 1
        torch.cuda.synchronize()
        return P, m, v
    Pa, ma, va = steps(ref_ctx, -1)
    ctx = ops.Context("cuda:0")   # fresh arena
    Pb, mb, vb = steps(ctx, 2)
    for k in Pa:
        assert torch.equal(Pa[k], Pb[k]), k
    assert torch.equal(ma, mb) and torch.equal(va, vb)
    ctx.close(); ref_ctx.close()


@pytest.mark.gpu
def test_view_chunks_give_the_gradients_of_the_whole_call():
    """A call above 2^31 tile intersections walks its views in chunks (st3r_gs_train_fwd_bwd; here forced by debug
    flag 32 on a small scene): loss, statistics and parameter gradients equal those of the one-pass call -- the loss
    is a sum over views (starster/gs.py:149-152); the float sums of the gradients associate differently, hence a
    tolerance -- and the chunk count sticks to the context."""
    import numpy as np
    from starst3r_amd import ops
    from st3r_synth import synth
    ctx = ops.Context("cuda:0")
    N, V, W, H = 30000, 5, 320, 240
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=11, scale_lo=0.004, scale_hi=0.03)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    P = {k: dev(v) for k, v in g.items()}
    rgb, _, _ = ops.render(ctx, P, vm, K, campos, W, H)
    gt = torch.clamp(rgb + 0.05 * torch.randn_like(rgb), 0, 1).contiguous()
    g1 = torch.empty(23 * N, device="cuda:0"); l1 = torch.zeros(1, device="cuda:0")
    st1 = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, g1, l1)
    ops.set_debug(ctx, 32)
    g2 = torch.full_like(g1, 7.0); l2 = torch.zeros(1, device="cuda:0")
    st2 = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, g2, l2)
    ops.set_debug(ctx, 0)
    assert st1["n_isects"] == st2["n_isects"] and st1["n_visible"] == st2["n_visible"]
    assert float(l1) == pytest.approx(float(l2), rel=1e-6)
    scale = g1.abs().max()
    assert float((g1 - g2).abs().max() / scale) < 1e-6
    # sticky: the next call of this context is chunked without the flag (and without statistics), same gradients
    g3 = torch.empty_like(g1)
    ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, g3, l2, want_stats=False)
    assert torch.equal(g2, g3)


@pytest.mark.gpu
@pytest.mark.parametrize("pruning", [False, True])
def test_run_3dgs_optim_repeats_an_iteration_whose_update_was_dropped(monkeypatch, pruning):
    """the ST3R_ERR_CAPACITY recovery of run_3dgs_optim itself (re-run iteration k-1, then k; the final
    ops.settle retry).  An asynchronous step is made to outgrow its buffers (debug flag 8 halves its capacity): its Adam
    update is skipped on the device, the next call reports it, the loop repeats the lost iteration.  Without the MCMC
    hooks the run equals an undisturbed one bit for bit; with them the position noise of the lost iteration was drawn before
    its (repeated) update instead of after it -- a documented deviation: same Gaussian count, finite losses, close parameters."""
    import starst3r_amd as st
    from starst3r_amd import gs as gs_mod, ops
    from st3r_synth.synth_model import SyntheticPairwiseModel

    def make_scene():
        scene = st.Scene(device="cuda:0")
        scene.add_images(SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2), [torch.zeros(3, 96, 128)] * 2)
        scene.init_3dgs()
        return scene
    iters = 7
    ref = make_scene()
    ref_losses = ref.run_3dgs_optim(iters, enable_pruning=pruning)
    real = ops.train_step
    calls = {"n": 0, "overflowed": []}

    def hooked(ctx, *a, **kw):
        k = calls["n"]; calls["n"] += 1
        if k in (3, iters):           # the 4th iteration, and -- counting the two repeats -- the LAST one (settle path)
            ops.set_debug(ctx, 8); calls["overflowed"].append(k)
        try:
            return real(ctx, *a, **kw)
        finally:
            ops.set_debug(ctx, 0)
    monkeypatch.setattr(gs_mod.ops, "train_step", hooked)
    sc = make_scene()
    losses = sc.run_3dgs_optim(iters, enable_pruning=pruning)
    # 7 iterations + the repeat of the 4th (two extra calls: k-1 and k again... k is the call that raised) + the final repeat
    assert calls["overflowed"] == [3, iters] and calls["n"] >= iters + 2
    assert len(losses) == iters and np.all(np.isfinite(losses))
    assert sc._gs_optim.step == ref._gs_optim.step == iters
    assert sc.gaussians["means"].shape == ref.gaussians["means"].shape
    if not pruning:
        for k in ("means", "quats", "scales", "opacities", "shN"):
            assert torch.equal(sc.gaussians[k].data, ref.gaussians[k].data), k
        assert losses == ref_losses
    else:
        for k in ("means", "scales", "opacities"):
            assert torch.allclose(sc.gaussians[k].data, ref.gaussians[k].data, atol=5e-3), k
        assert abs(losses[-1] - ref_losses[-1]) < 0.05 * abs(ref_losses[-1])


def test_release_scratch_frees_the_arena_and_training_goes_on_unchanged():
    """st3r_ctx_release_scratch between two steps: the arena is empty afterwards, the context stays valid, and the run
    continues bit-identically to an uninterrupted one (asynchronous steps, i.e. with the record-count hint in play; the
    stamped gradient slots, the scan's status words and the count words are re-initialised when they are allocated again)."""
    import numpy as np
    from starst3r_amd import ops
    from st3r_synth import synth
    N, V, W, H = 20000, 3, 320, 240
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=6, scale_lo=0.004, scale_hi=0.03)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    P0 = {k: dev(v) for k, v in g.items()}

    def run(release_at):
        ctx = ops.Context("cuda:0")
        rgb, _, _ = ops.render(ctx, P0, vm, K, campos, W, H)
        gt = torch.clamp(rgb + 0.05, 0, 1).contiguous()
        P = {k: v.clone() for k, v in P0.items()}
        grads = torch.empty(23 * N, device="cuda:0"); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        losses = torch.zeros(6, device="cuda:0")
        for it in range(6):
            if it in release_at:
                assert ctx.arena_bytes() > 0
                ctx.release_scratch()
                assert ctx.arena_bytes() == 0
            ops.train_step(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                           losses[it:it + 1], want_stats=it == 0)
        torch.cuda.synchronize()
        ctx.close()
        return P, losses
    Pa, la = run(())
    Pb, lb = run((2, 5))
    assert torch.equal(la, lb)
    for k in Pa:
        assert torch.equal(Pa[k], Pb[k]), k
