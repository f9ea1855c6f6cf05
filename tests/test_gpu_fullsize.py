"""BASELINE.json's full size (SYNTH-1M: 1 M Gaussians, 8 x 1920x1080 views) through size-independent
properties -- the CPU oracle cannot finish this in seconds, the invariants can be checked on the device:

  * integer pipeline: intersection count = sum of tile counts (a checksum of checksums), keys sorted, offsets =
    run boundaries of the sorted keys, values a permutation of the emitted pairs, the fused path's two-level sort
    reproduces the single 64-bit sort order exactly (stable ties included);
  * blending: alpha in [0, 1), rgb finite and >= 0, last_ids inside their tile's list, render == fused path's image;
  * backward: linear in v_rgb (superposition), zero for zero input;
  * fused train step: gradients equal the staged path's (exact culling drops only (record, tile) pairs that fail
    the alpha test everywhere), Adam equals torch.optim.Adam on 23 M scalars, loss decreases;
  * loss: SSIM(x, x) = 1, L1 symmetric;
  * MCMC hooks at 1 M: every dead Gaussian relocated, draw counts sum to the number of dead.
All through the C ABI (starst3r_amd.ops), one MI355X.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from st3r_synth import synth

DEV = "cuda:0"
N, V, W, H = 1_000_000, 8, 1920, 1080


@pytest.fixture(scope="module")
def S():
    from starst3r_amd import ops
    ctx = ops.get_context(DEV)
    g, w2c, Ks = synth.make_scene(N, V, W, H)
    P = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    w2c = torch.tensor(w2c, device=DEV); Ks = torch.tensor(Ks, device=DEV)
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], w2c, Ks, W, H)
    torch.cuda.synchronize()
    return dict(ctx=ctx, P=P, w2c=w2c, Ks=Ks, rgb=rgb, alpha=alpha, info=info, campos=ops.camera_positions(w2c))


def test_integer_pipeline_invariants(S):
    info = S["info"]
    ids = info["isect_ids"]; off = info["isect_offsets"].reshape(-1).long()
    I = ids.numel()
    assert I == int(info["tiles_per_gauss"].long().sum())                 # checksum of checksums
    assert I > 30_000_000 and I == int(info["_cum_tiles"][-1])
    assert bool((ids[1:] >= ids[:-1]).all())                              # sorted
    tw, th = info["tile_width"], info["tile_height"]
    tb = (tw * th).bit_length()
    tile_of = ((ids >> 32) & ((1 << tb) - 1)) + (ids >> (32 + tb)) * (tw * th)    # linear (camera, tile) id
    assert int(tile_of.max()) < V * tw * th
    # offsets[k] = first position whose (camera, tile) id is >= k
    expect = torch.searchsorted(tile_of, torch.arange(V * tw * th, device=DEV))
    assert torch.equal(off, expect)
    # values: every visible pair appears exactly tiles_per_gauss times
    cnt = torch.bincount(info["_flatten_ids_dense"].long(), minlength=V * N)
    dense_tiles = torch.zeros(V * N, dtype=torch.int64, device=DEV)
    pid = (info["camera_ids"].long() * N + info["gaussian_ids"].long())
    dense_tiles[pid] = info["tiles_per_gauss"].long()
    assert torch.equal(cnt, dense_tiles)
    # depth bits of the key equal the depth of the pair they point to
    d = S["info"]["_splats"][info["_flatten_ids_dense"].long(), 9].view(torch.int32).long()
    assert torch.equal(ids & 0xFFFFFFFF, d)


def test_two_level_sort_equals_single_sort_at_full_size(S):
    """The fused path sorts (camera | depth) pairs, then (camera, tile) records stably; restricted to the records it
    keeps, the order must be the single 64-bit sort's order."""
    from starst3r_amd import ops
    ctx, P = S["ctx"], S["P"]
    gt = torch.zeros((V, H, W, 3), device=DEV)
    grads = torch.empty(23 * N, device=DEV); loss = torch.zeros(1, device=DEV)
    st = ops.train_fwd_bwd(ctx, P, S["w2c"], S["Ks"], S["campos"], gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    kept = st["n_isects"]
    assert st["n_isects_ref"] == S["info"]["isect_ids"].numel() and 0 < kept < st["n_isects_ref"]
    fused_flat = ops.peek(ctx, 0, kept)                                   # sorted pair ids of the fused path
    fused_off = ops.peek(ctx, 1, V * S["info"]["tile_width"] * S["info"]["tile_height"]).long()
    ref_flat = S["info"]["_flatten_ids_dense"]; ref_off = S["info"]["isect_offsets"].reshape(-1).long()
    # per tile the fused list must be a subsequence of the reference list, in the same order: check by
    # walking both with a vectorised merge -- position of each fused record inside the reference list
    tile_of_fused = torch.searchsorted(fused_off, torch.arange(kept, device=DEV), right=True) - 1
    key_ref = (torch.searchsorted(ref_off, torch.arange(ref_flat.numel(), device=DEV), right=True) - 1) * (V * N) \
        + ref_flat.long()
    key_fused = tile_of_fused * (V * N) + fused_flat.long()
    # every (tile, pair) of the fused path exists in the reference path ...
    srt, order = torch.sort(key_ref)
    pos = torch.searchsorted(srt, key_fused)
    assert bool((srt[pos.clamp(max=srt.numel() - 1)] == key_fused).all())
    # ... and their reference positions increase along the fused list (same relative order, ties included)
    ref_pos = order[pos]
    assert bool((ref_pos[1:] > ref_pos[:-1]).all())


def test_blend_properties_and_render_equals_fused_image(S):
    from starst3r_amd import ops
    rgb, alpha, info = S["rgb"], S["alpha"], S["info"]
    assert bool(torch.isfinite(rgb).all()) and float(rgb.min()) >= 0.0
    assert float(alpha.min()) >= 0.0 and float(alpha.max()) < 1.0
    assert float(alpha.mean()) > 0.3                                       # the scene covers the views
    last = info["_last_ids"].reshape(V, H, W)
    off = info["isect_offsets"]
    I = info["isect_ids"].numel()
    ty = torch.arange(H, device=DEV) // 16; tx = torch.arange(W, device=DEV) // 16
    start = off[:, ty][:, :, tx].long()
    nxt = torch.cat([off.reshape(-1)[1:], torch.tensor([I], device=DEV, dtype=off.dtype)]).reshape(off.shape)
    end = nxt[:, ty][:, :, tx].long()
    touched = alpha.reshape(V, H, W) > 0
    assert bool(((last >= start) & (last < end))[touched].all())
    # st3r_gs_render (one call, ctx scratch) produces the same image as the staged path
    r2, a2, _ = ops.render(S["ctx"], S["P"], S["w2c"], S["Ks"], S["campos"], W, H)
    assert torch.equal(r2, rgb) and torch.equal(a2, alpha)


def test_culling_changes_nothing_at_full_size(S):
    """Both culling levels only drop work whose alpha is below 1/255 everywhere, so the images must be bit-identical:
    (a) per-quadrant relevance test off vs on (st3r_ctx_set_debug), (b) the fused path's exact tile culling vs the
    reference rectangles of the staged path."""
    from starst3r_amd import ops
    ctx = S["ctx"]
    ops.set_debug(ctx, 1)
    try:
        r_all, a_all, _ = ops.render(ctx, S["P"], S["w2c"], S["Ks"], S["campos"], W, H)
    finally:
        ops.set_debug(ctx, 0)
    assert torch.equal(r_all, S["rgb"]) and torch.equal(a_all, S["alpha"])
    # (c) the cell-list kernel of the training path (flag 512), with its exact cell test and without (513)
    for flags in (512, 513):
        ops.set_debug(ctx, flags)
        try:
            r_c, a_c, _ = ops.render(ctx, S["P"], S["w2c"], S["Ks"], S["campos"], W, H)
        finally:
            ops.set_debug(ctx, 0)
        assert torch.equal(r_c, S["rgb"]) and torch.equal(a_c, S["alpha"]), flags
    gt = torch.zeros((V, H, W, 3), device=DEV)
    grads = torch.empty(23 * N, device=DEV); loss = torch.zeros(1, device=DEV)
    st = ops.train_fwd_bwd(ctx, S["P"], S["w2c"], S["Ks"], S["campos"], gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    assert st["n_isects"] < st["n_isects_ref"]
    fused_rgb = ops.peek(ctx, 8, V * H * W * 3, torch.float32).reshape(V, H, W, 3)
    fused_alpha = ops.peek(ctx, 9, V * H * W, torch.float32).reshape(V, H, W, 1)
    assert torch.equal(fused_rgb, S["rgb"]) and torch.equal(fused_alpha, S["alpha"])


def test_backward_is_linear_in_v_rgb(S):
    from starst3r_amd import ops
    ctx, info = S["ctx"], S["info"]
    gen = torch.Generator(device=DEV).manual_seed(3)
    v1 = torch.randn((V, H, W, 3), device=DEV, generator=gen); v2 = torch.randn((V, H, W, 3), device=DEV, generator=gen)

    def bwd(v):
        ops.blend_fwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], V, W, H)
        return ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], S["alpha"],
                             info["_last_ids"], v, None, info["_cum_tiles"], V, W, H)
    g1, g2, g12 = bwd(v1), bwd(v2), bwd(v1 + 2.0 * v2)
    scale = float(g12.abs().max())
    assert scale > 0
    err = float((g12 - (g1 + 2.0 * g2)).abs().max())
    assert err <= 2e-4 * scale, (err, scale)                               # superposition, fp32 summation order only
    assert float(bwd(torch.zeros_like(v1)).abs().max()) == 0.0


def test_fused_gradients_equal_staged_path_and_adam_equals_torch(S):
    from starst3r_amd import ops
    ctx, P, info = S["ctx"], S["P"], S["info"]
    gt_np = synth.perturb_for_gt({k: v.cpu().numpy() for k, v in P.items()})
    Q = {k: torch.tensor(v, device=DEV) for k, v in gt_np.items()}
    gt, _, _ = ops.render(ctx, Q, S["w2c"], S["Ks"], S["campos"], W, H)
    gt = gt.clamp(0, 1).contiguous()
    grads = torch.empty(23 * N, device=DEV); loss = torch.zeros(1, device=DEV)
    ops.train_fwd_bwd(ctx, P, S["w2c"], S["Ks"], S["campos"], gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    # staged path: loss kernel + blend backward + projection backward on the reference rectangles
    sums, v_rgb = ops.loss_l1_ssim(ctx, S["rgb"], gt, 0.8, 0.2, want_grad=True)
    ops.blend_fwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], V, W, H)
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], S["alpha"],
                             info["_last_ids"], v_rgb, None, info["_cum_tiles"], V, W, H)
    ref = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], S["w2c"], S["Ks"],
                             S["campos"], W, H, info["_splats"], v_splats, reg_views=float(V), opac_fac=0.01,
                             scale_fac=0.01)
    off = 0
    for name, wdt in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1), ("sh", 12)):
        a = grads[off * N:(off + wdt) * N]; b = ref[off * N:(off + wdt) * N]; off += wdt
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-12, name
    l_ref = float((0.8 * sums[:, 0] / (H * W * 3) + 0.2 * (1 - sums[:, 1] / ((H - 10) * (W - 10) * 3))).sum())
    reg = V * (0.01 * float(torch.sigmoid(P["opacities"]).mean()) + 0.01 * float(torch.exp(P["scales"]).mean()))
    assert float(loss) == pytest.approx(l_ref + reg, rel=1e-5)
    # Adam on all 23 M scalars vs torch.optim.Adam (single tensor semantics)
    A = {k: v.clone() for k, v in P.items()}
    m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    flat = torch.cat([P["means"].reshape(-1), P["quats"].reshape(-1), P["scales"].reshape(-1), P["opacities"].reshape(-1),
                      P["shN"][:, :4].reshape(-1)]).clone().requires_grad_(True)
    opt = torch.optim.Adam([flat], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
    for step in (1, 2):
        ops.adam_step(ctx, A, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        flat.grad = grads.clone(); opt.step()
    got = torch.cat([A["means"].reshape(-1), A["quats"].reshape(-1), A["scales"].reshape(-1), A["opacities"].reshape(-1),
                     A["shN"][:, :4].reshape(-1)])
    assert float((got - flat.detach()).abs().max()) <= 2e-7
    assert torch.equal(A["shN"][:, 4:], P["shN"][:, 4:])                   # unused SH rows untouched


def test_training_reduces_the_loss_at_full_size(S):
    from starst3r_amd import ops
    ctx = S["ctx"]
    P = {k: v.clone() for k, v in S["P"].items()}
    gt_np = synth.perturb_for_gt({k: v.cpu().numpy() for k, v in P.items()})
    Q = {k: torch.tensor(v, device=DEV) for k, v in gt_np.items()}
    gt, _, _ = ops.render(ctx, Q, S["w2c"], S["Ks"], S["campos"], W, H)
    gt = gt.clamp(0, 1).contiguous()
    grads = torch.empty(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    losses = torch.zeros(12, device=DEV)
    for it in range(12):
        ops.train_step(ctx, P, S["w2c"], S["Ks"], S["campos"], gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999,
                       1e-8, it + 1, losses[it:it + 1])
    L = losses.cpu().numpy()
    assert np.isfinite(L).all() and L[-1] < 0.8 * L[0], L


def test_loss_identities_at_full_size(S):
    from starst3r_amd import ops
    x = S["rgb"]
    sums, _ = ops.loss_l1_ssim(S["ctx"], x, x, 0.8, 0.2, want_grad=False)
    assert float(sums[:, 0].abs().max()) == 0.0
    np.testing.assert_allclose((sums[:, 1] / ((H - 10) * (W - 10) * 3)).cpu().numpy(), 1.0, rtol=0, atol=1e-6)
    y = x.flip(0).contiguous()
    a, _ = ops.loss_l1_ssim(S["ctx"], x, y, 1.0, 0.0, want_grad=False)
    b, _ = ops.loss_l1_ssim(S["ctx"], y, x, 1.0, 0.0, want_grad=False)
    np.testing.assert_allclose(a[:, 0].cpu().numpy(), b[:, 0].cpu().numpy(), rtol=1e-12)


def test_mcmc_relocation_at_full_size(S):
    from starst3r_amd import ops
    ctx = S["ctx"]
    P = {k: v.clone() for k, v in S["P"].items()}
    P["sh0"] = torch.zeros(N, 1, 3, device=DEV)
    P["opacities"][::17] = -8.0
    n_dead = int((torch.sigmoid(P["opacities"]) <= 0.005).sum())
    m = torch.ones(23 * N, device=DEV); v = torch.ones(23 * N, device=DEV)
    assert ops.mcmc_relocate(ctx, P, m, v, 0.005, seed=11, step=0) == n_dead
    counts = ops.peek(ctx, 7, N)
    assert int(counts.sum()) == n_dead
    assert int((torch.sigmoid(P["opacities"]) <= 0.005 - 1e-6).sum()) == 0
    sources = counts > 0
    assert float(m[:3 * N].reshape(N, 3)[sources].abs().max()) == 0.0 and float(m[:3 * N].reshape(N, 3)[~sources].min()) == 1.0


def test_larger_than_baseline_config_runs():
    """Towards BASELINE configs[4] (5 M Gaussians, 4K views, 8 per GPU): 3 M Gaussians x 4 views of 3840x2160 on one
    GPU -- 18-bit tile keys, ~10^8 intersections, ~10 GB of scratch -- trains and the loss goes down."""
    from starst3r_amd import ops
    ctx = ops.get_context(DEV)
    n, v, w, h = 3_000_000, 4, 3840, 2160
    g, w2c_np, Ks_np = synth.make_scene(n, v, w, h)
    P = {k: torch.tensor(val, device=DEV) for k, val in g.items()}
    w2c = torch.tensor(w2c_np, device=DEV); Ks = torch.tensor(Ks_np, device=DEV)
    campos = ops.camera_positions(w2c)
    Q = {k: torch.tensor(val, device=DEV) for k, val in synth.perturb_for_gt(g).items()}
    gt, _, _ = ops.render(ctx, Q, w2c, Ks, campos, w, h)
    gt = gt.clamp(0, 1).contiguous()
    del Q
    grads = torch.empty(23 * n, device=DEV); m = torch.zeros_like(grads); vv = torch.zeros_like(grads)
    losses = torch.zeros(6, device=DEV)
    st = None
    for it in range(6):
        st = ops.train_step(ctx, P, w2c, Ks, campos, gt, w, h, 0.2, 0.01, 0.01, grads, m, vv, 1e-3, 0.9, 0.999, 1e-8,
                            it + 1, losses[it:it + 1])
    L = losses.cpu().numpy()
    assert st["n_isects_ref"] > 80_000_000 and st["n_isects"] < st["n_isects_ref"]
    assert np.isfinite(L).all() and L[-1] < L[0]
    assert bool(torch.isfinite(grads).all())
