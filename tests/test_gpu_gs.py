"""GPU parity tests for path C: every HIP stage vs oracle/gs_oracle.c through the C ABI.

Integer outputs (radii, ids, tile counts, sort keys/values, offsets) must be BIT-EXACT;
float outputs within the tolerances written next to each assert.  Run on the MI355X box:
    python -m pytest tests -m gpu -x -q
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as go
from st3r_synth import synth


@pytest.fixture(scope="module")
def ctx():
    from starst3r_amd import ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return ops.get_context("cuda:0")


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


SCENES = {
    # name: (N, views, W, H, seed, scale_lo, scale_hi)
    "small": (400, 3, 96, 64, 7, 0.01, 0.08),
    "ragged": (1500, 2, 101, 75, 21, 0.01, 0.12),     # image not a multiple of the tile size
    "medium": (20000, 4, 320, 240, 5, 0.004, 0.03),
    "one": (1, 1, 48, 32, 1, 0.05, 0.06),
    "many": (800, 9, 64, 48, 3, 0.01, 0.08),          # > 8 views: 64-bit level-1 keys, row-interleaved XCD map
    # large Gaussians, ~25 tiles per visible pair (the regime of the configs[1] example after a few hundred iterations): the
    # projection backward's slot gather takes its item-parallel form (gs_project_bwd.hip, round 6)
    "wide": (2500, 3, 320, 240, 13, 0.08, 0.25),
}


def fuzz_scene(seed):
    """Randomised stress scene: sub-pixel to image-filling anisotropic Gaussians, means also behind / beside / right in
    front of the cameras, raw opacities outside [0, 1] (the reference renders them raw), unnormalised quaternions,
    odd image sizes."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(200, 1200)); V = int(rng.integers(1, 5))
    W = int(rng.integers(40, 200)); H = int(rng.integers(30, 150))
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=seed, scale_lo=1e-3, scale_hi=0.5)
    g["means"] = rng.uniform(-3.6, 3.6, (N, 3)).astype(np.float32)
    g["opacities"] = rng.uniform(-0.3, 1.6, N).astype(np.float32)
    g["scales"] = (g["scales"] * rng.uniform(0.2, 5.0, (N, 3))).astype(np.float32)
    g["quats"] = (g["quats"] * rng.uniform(0.1, 3.0, (N, 1))).astype(np.float32)
    return g, w2c, Ks, W, H


FUZZ = ["fuzz%d" % i for i in range(12)]
# fraction of pixels whose skip / stop decisions float32 determines (oracle margin > 1e-4), per stress scene
FUZZ_CHECKED = {"fuzz0": 0.9207, "fuzz1": 0.8785, "fuzz2": 0.6398, "fuzz3": 0.9991, "fuzz4": 0.9981, "fuzz5": 0.9680,
                "fuzz6": 0.9923, "fuzz7": 0.9416, "fuzz8": 0.9893, "fuzz9": 0.9939, "fuzz10": 0.9666, "fuzz11": 0.9026}


def make(name):
    if name.startswith("fuzz"):
        return fuzz_scene(int(name[4:]))
    N, V, W, H, seed, lo, hi = SCENES[name]
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=seed, scale_lo=lo, scale_hi=hi)
    return g, w2c, Ks, W, H


def run_hip(ctx, g, w2c, Ks, W, H):
    from starst3r_amd import ops
    P = {k: dev(v) for k, v in g.items()}
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], dev(w2c),
                                         dev(Ks), W, H)
    torch.cuda.synchronize()
    return P, rgb, alpha, info


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "one", "wide"] + FUZZ)
def test_projection_tiles_sort_offsets_bit_exact(ctx, name):
    g, w2c, Ks, W, H = make(name)
    _, _, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
    P, rgb, alpha, info = run_hip(ctx, g, w2c, Ks, W, H)
    n = lambda t: t.cpu().numpy()
    for key in ("camera_ids", "gaussian_ids", "radii", "tiles_per_gauss"):
        assert np.array_equal(n(info[key]), meta[key]), key
    # same IEEE op sequence on both sides -> identical bits for the projected geometry
    for key in ("means2d", "depths", "conics"):
        assert np.array_equal(n(info[key]).view(np.uint32), meta[key].view(np.uint32)), key
    np.testing.assert_allclose(n(info["colors"]), meta["colors"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(n(info["opacities"]), meta["opacities"])
    assert np.array_equal(n(info["isect_ids"]), meta["isect_ids"])
    assert np.array_equal(n(info["flatten_ids"]), meta["flatten_ids"])
    assert np.array_equal(n(info["isect_offsets"]), meta["isect_offsets"])
    assert np.array_equal(n(info["_isect_ids_unsorted"]), meta["isect_ids_unsorted"])
    assert meta["isect_ids"].size > 0


def test_radius_constant_kat(ctx):
    """Known answer from the formula itself (radius = ceil(3 sqrt(b + sqrt(max(0.01, b^2 - det))))),
    not from the oracle: isotropic cov2d 0.8 -> radius 3 (the INRIA constant 0.1 would give 4)."""
    from kat_scenes import radius_kat_scene
    g, V, K, W, H = radius_kat_scene()
    P, rgb, alpha, info = run_hip(ctx, g, V, K, W, H)
    assert info["radii"].cpu().numpy().tolist() == [3]
    assert info["tiles_per_gauss"].cpu().numpy().tolist() == [1]
    assert info["isect_ids"].numel() == 1


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "one", "many", "wide"] + FUZZ)
def test_fused_two_level_sort_matches_reference_order(ctx, name):
    """The fused render/train path sorts in two levels ((camera|depth) then a stable (camera,tile)
    pass); its sorted pair ids and tile offsets must equal the oracle's single 64-bit-key sort."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks,
                                            W, H, want_margin=True)
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    rgb, alpha, st = ops.render(ctx, P, vm, K, ops.camera_positions(vm), W, H)
    n = st["n_isects"]
    assert n == meta["isect_ids"].size
    N = g["means"].shape[0]
    flat_dense = ops.peek(ctx, 0, n).cpu().numpy()
    off = ops.peek(ctx, 1, meta["isect_offsets"].size).cpu().numpy()
    # oracle flatten ids are packed indices: map to dense pair ids cam*N + gaussian
    dense_of_packed = meta["camera_ids"].astype(np.int64) * N + meta["gaussian_ids"]
    assert np.array_equal(flat_dense, dense_of_packed[meta["flatten_ids"]])
    assert np.array_equal(off, meta["isect_offsets"].reshape(-1))
    ok = meta["margin"] > 1e-4
    np.testing.assert_allclose(rgb.cpu().numpy()[ok], rgb_o[ok], rtol=1e-4, atol=1e-5)


def _kept_list_against_oracle(ctx, g, w2c, Ks, W, H, meta, n_kept, max_dropped=2_000_000):
    """The TRAINING path's record list (exact culling: a (record, tile) pair is kept only if the box of {alpha >= 1/255}
    reaches the tile's pixel centres) against the oracle's sorted list (gsplat's 3-sigma squares):
      * kept list == the oracle's list with the dropped (record, tile) pairs removed, ORDER INCLUDED (north_star: tile / sort
        indices bit-exact);
      * tile offsets == the run boundaries of that filtered list;
      * every dropped pair fails the alpha test on all 256 pixels of its tile (float64 evaluation of the oracle's formula).
    Returns (#oracle pairs, #dropped)."""
    from starst3r_amd import ops
    N, Cn = g["means"].shape[0], w2c.shape[0]
    tw, th = (W + 15) // 16, (H + 15) // 16
    n_tiles = tw * th
    kept_pid = ops.peek(ctx, 0, n_kept).cpu().numpy().astype(np.int64)
    off = ops.peek(ctx, 1, Cn * n_tiles + 1).cpu().numpy().astype(np.int64)
    assert off[-1] == n_kept and np.all(np.diff(off) >= 0)
    kept_tile = np.repeat(np.arange(Cn * n_tiles, dtype=np.int64), np.diff(off))        # camera * tiles + tile
    # the oracle's sorted list: key = cam << (32 + tile_bits) | tile << 32 | depth bits, value = PACKED index
    tile_bits = int(n_tiles).bit_length()
    keys = meta["isect_ids"].astype(np.int64)
    o_tile = ((keys >> 32) & ((1 << tile_bits) - 1)) + (keys >> (32 + tile_bits)) * n_tiles
    packed = meta["flatten_ids"].astype(np.int64)
    o_pid = meta["camera_ids"].astype(np.int64)[packed] * N + meta["gaussian_ids"].astype(np.int64)[packed]
    code = lambda tile, pid: tile * (N * Cn) + pid                                        # one integer per (tile, pair)
    kept_code, o_code = code(kept_tile, kept_pid), code(o_tile, o_pid)
    keep = np.isin(o_code, kept_code)
    assert keep.sum() == n_kept, "a kept pair that the reference algorithm does not have"
    assert np.array_equal(o_code[keep], kept_code), "the kept list is not the oracle's list minus the dropped pairs"
    # dropped pairs: alpha < 1/255 on every pixel centre of the tile
    d_tile, d_packed = o_tile[~keep], packed[~keep]
    if d_tile.size > max_dropped:
        sel = np.random.default_rng(0).choice(d_tile.size, max_dropped, replace=False)
        d_tile, d_packed = d_tile[sel], d_packed[sel]
    t = d_tile % n_tiles
    px0 = (t % tw) * 16 + 0.5; py0 = (t // tw) * 16 + 0.5
    m2 = meta["means2d"].astype(np.float64)[d_packed]; con = meta["conics"].astype(np.float64)[d_packed]
    opac = meta["opacities"].astype(np.float64)[d_packed]
    worst = 0.0
    for s0 in range(0, d_tile.size, 50_000):
        sl = slice(s0, s0 + 50_000)
        dx = m2[sl, 0, None, None] - (px0[sl, None, None] + np.arange(16)[None, None, :])
        dy = m2[sl, 1, None, None] - (py0[sl, None, None] + np.arange(16)[None, :, None])
        sigma = 0.5 * (con[sl, 0, None, None] * dx * dx + con[sl, 2, None, None] * dy * dy) + con[sl, 1, None, None] * dx * dy
        alpha = np.minimum(0.999, opac[sl, None, None] * np.exp(-sigma))
        alpha[sigma < 0] = 0.0                                                            # gsplat skips sigma < 0
        worst = max(worst, float(alpha.max()) if alpha.size else 0.0)
    assert worst < 1.0 / 255.0, worst
    return o_code.size, int((~keep).sum())


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "one", "many", "wide"] + FUZZ)
def test_fused_training_list_is_the_oracle_list_minus_dead_pairs(ctx, name):
    """the timed (fused training) path's record list was only ever compared with the oracle indirectly (same
    images, gradients close to the staged path's).  Directly: see _kept_list_against_oracle."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    N, Cn = g["means"].shape[0], w2c.shape[0]
    rgb_o, _, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    gt = dev(np.clip(rgb_o, 0, 1))
    grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    st = ops.train_fwd_bwd(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    torch.cuda.synchronize()
    assert st["n_isects_ref"] == meta["isect_ids"].size
    n_ref, n_drop = _kept_list_against_oracle(ctx, g, w2c, Ks, W, H, meta, st["n_isects"])
    assert n_ref - n_drop == st["n_isects"]


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "wide"] + FUZZ)
def test_blend_forward(ctx, name):
    g, w2c, Ks, W, H = make(name)
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks,
                                            W, H, want_margin=True)
    P, rgb, alpha, info = run_hip(ctx, g, w2c, Ks, W, H)
    rgb, alpha, last = rgb.cpu().numpy(), alpha.cpu().numpy(), info["_last_ids"].cpu().numpy()
    # pixels whose skip/stop decisions sit within 1e-4 (relative) of a threshold may legally
    # flip under a 1-ulp exp difference (v_exp_f32 vs glibc expf): excluded, and they must be rare
    ok = meta["margin"] > 1e-4
    # (the randomised stress scenes are full of strongly anisotropic, image-filling Gaussians whose sigma is not
    # determined to 1e-4 in float32 far from the mean: more pixels are excluded there, see gso_blend_fwd).  The
    # excluded fraction is a property of the scene and of the oracle alone (CPU, deterministic), so it is pinned per
    # scene instead of bounded by one loose number: 9 of the 12 stress scenes keep > 90 % of their pixels
    # ("wide": ~100 large Gaussians deep per pixel, i.e. ~100 decisions per pixel that can sit near a threshold: 0.9974)
    floor = FUZZ_CHECKED[name] - 2e-3 if name.startswith("fuzz") else (0.997 if name == "wide" else 0.999)
    assert ok.mean() >= floor, (name, ok.mean())
    if name.startswith("fuzz") or name == "wide":
        # the excluded pixels are not left unchecked: whichever way their borderline decisions fall, the image stays
        # within about one 1/255 step of the oracle's
        bad = ~ok
        if bad.any():
            worst = float(np.abs(rgb[bad] - rgb_o[bad]).max())
            print(name, "checked fraction %.4f, worst |rgb - oracle| on the excluded pixels %.3e" % (ok.mean(), worst))
            assert worst <= 5e-3, (name, worst)   # measured: 1e-7 .. 1.4e-3
    # tolerance: 1e-4 relative (north_star) with a 1e-5 absolute floor for near-zero pixels
    np.testing.assert_allclose(rgb[ok], rgb_o[ok], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(alpha[ok], alpha_o[ok], rtol=1e-4, atol=1e-5)
    assert np.array_equal(last[ok], meta["last_ids"][ok])
    assert alpha_o.max() > 0.3


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "wide"] + FUZZ)
def test_backward_vs_oracle(ctx, name):
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks,
                                            W, H, want_margin=True)
    rng = np.random.default_rng(3)
    v_rgb = rng.standard_normal(rgb_o.shape).astype(np.float32)
    v_alpha = rng.standard_normal(alpha_o.shape).astype(np.float32)
    # no gradient enters through pixels that float32 does not determine (decision within 1e-4 of a threshold, or
    # sigma lost to cancellation): their control flow may legitimately differ between oracle and kernel
    und = ~(meta["margin"] > 1e-4)
    v_rgb[und] = 0.0; v_alpha[und] = 0.0
    go_grads = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H,
                                         meta, alpha_o, v_rgb, v_alpha)
    P, rgb, alpha, info = run_hip(ctx, g, w2c, Ks, W, H)
    Cn, N = w2c.shape[0], g["means"].shape[0]
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                             info["_last_ids"], dev(v_rgb), dev(v_alpha), info["_cum_tiles"], Cn, W, H)
    torch.cuda.synchronize()
    # per-pair gradients: compare in packed order
    pid = (info["camera_ids"].long() * N + info["gaussian_ids"].long())
    vs = v_splats[pid].cpu().numpy()
    pk = go_grads["packed"]

    dist = {}
    maxerr = {}
    # measured (round 3): regular scenes median <= 7e-7, p99 <= 4e-5; stress scenes median <= 4e-5, p99 <= 5e-3
    # ("wide": a record's gradient is a float32 sum over ~25 tiles x up to 256 pixels, grouped differently in kernel and oracle:
    # median 1e-6 like the regular scenes, p99 1.1e-4)
    MED_BOUND, P99_BOUND = (1e-4, 1e-2) if name.startswith("fuzz") else ((2e-6, 3e-4) if name == "wide" else (2e-6, 1e-4))

    # bounds against the tensor's maximum, set from the measured errors (round 4: regular scenes <= 1.5e-5 per pair and
    # <= 1.0e-5 per parameter, stress scenes <= 6.7e-5 and <= 1.0e-3 -- the needle-shaped Gaussians' quaternion chain rule);
    # round 3 allowed 2e-4 / 1e-3 / 4e-3
    PAIR_TOL, PARAM_TOL = (2e-4, 3e-3) if name.startswith("fuzz") else (5e-5, 5e-5)

    def close(a, b, name, tol=PAIR_TOL):
        # the bound: relative to the tensor's max magnitude (the sums group differently in kernel and oracle) ...
        scale = np.abs(b).max() + 1e-20
        err = np.abs(a - b).max() / scale
        maxerr[name] = err
        assert err < tol, (name, err)
        # ... which says nothing about small-gradient Gaussians, so the DISTRIBUTION of the element-wise relative error
        # is pinned as well, over the elements above 1e-4 of the maximum (below that the float32 sums are rounding noise)
        big = np.abs(b) > 1e-4 * scale
        if big.sum() >= 50:
            rel = np.abs(a - b)[big] / np.abs(b)[big]
            med, p99 = float(np.median(rel)), float(np.percentile(rel, 99))
            dist[name] = (med, p99)
            assert med <= MED_BOUND and p99 <= P99_BOUND, (name, med, p99)
    close(vs[:, 0:2], pk["v_means2d"], "v_means2d")
    close(vs[:, 2], pk["v_opacities"], "v_opacities")
    close(vs[:, 3:6], pk["v_conics"], "v_conics")
    close(vs[:, 6:9], pk["v_colors"], "v_colors")
    grads = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], dev(w2c), dev(Ks),
                               info["_campos"], W, H, info["_splats"], v_splats)
    torch.cuda.synchronize()
    G = {k: v.cpu().numpy() for k, v in ops.split_grads(grads, N).items()}
    # (stress scenes: needle-shaped Gaussians make the covariance chain rule sum terms far larger than the result)
    for k in ("means", "quats", "scales", "opacities", "sh"):
        close(G[k], go_grads[k], k, tol=PARAM_TOL)
    print(name, "relative gradient error (median, p99):", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in dist.items()})
    print(name, "max error / tensor max:", {k: f"{v:.1e}" for k, v in maxerr.items()})


def test_backward_is_deterministic(ctx):
    """The backward has no atomics (per-wave accumulator rows, stamped slots, in-order gather): two runs agree bit
    for bit."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make("medium")
    P, rgb, alpha, info = run_hip(ctx, g, w2c, Ks, W, H)
    v_rgb = torch.randn(rgb.shape, device=rgb.device, generator=torch.Generator(device=rgb.device).manual_seed(5))
    Cn = w2c.shape[0]
    a = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                      info["_last_ids"], v_rgb, None, info["_cum_tiles"], Cn, W, H)
    b = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                      info["_last_ids"], v_rgb, None, info["_cum_tiles"], Cn, W, H)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("shape", [(1, 30, 37), (2, 64, 96), (1, 75, 101)])
def test_l1_ssim_vs_oracle(ctx, shape):
    from starst3r_amd import ops
    Cn, H, W = shape
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 1, (Cn, H, W, 3)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    sums, v = ops.loss_l1_ssim(ctx, dev(x), dev(y), 0.8, 0.2)
    torch.cuda.synchronize()
    sums = sums.cpu().numpy(); v = v.cpu().numpy()
    for c in range(Cn):
        l1, ss, vr = go.l1_ssim(x[c], y[c], 0.8, 0.2)
        assert abs(sums[c, 0] / (H * W * 3) - l1) < 1e-6
        assert abs(sums[c, 1] / ((H - 10) * (W - 10) * 3) - ss) < 1e-5
        # v_render (round 3: one rtol = 2e-3 over everything): the error against the tensor's maximum, and element-wise
        # over the elements that are not cancellation residue (above 1e-3 of the maximum).  Measured: 6e-7 / 1.5e-6
        # (v_rcp_f32 instead of IEEE divisions is the largest contribution); bounds 2e-6 / 1e-5
        scale = np.abs(vr).max()
        err = np.abs(v[c] - vr)
        big = np.abs(vr) > 1e-3 * scale
        assert err.max() <= 2e-6 * scale
        assert (err[big] / np.abs(vr[big])).max() <= 1e-5


@pytest.mark.parametrize("shape", [(1, 30, 37), (3, 64, 96), (2, 75, 101), (1, 200, 333)])
def test_gt_moments_change_no_bit_of_the_loss(ctx, shape):
    """Round 6: conv(gt) and conv(gt^2) are computed once per training call (st3r_loss_gt_moments) and read by the fused loss
    kernel instead of being convolved every iteration.  Same taps in the same order: the sums and the gradient image are the
    same BITS with and without the registered moments -- for the whole image set and for a whole-view offset into it (view
    shards / view chunks) -- and the moments equal a float64 convolution of the ground truth (starster/gs.py:129: torchmetrics
    recomputes them every call)."""
    from starst3r_amd import ops
    Cn, H, W = shape
    rng = np.random.default_rng(5)
    x = dev(rng.uniform(0, 1, (Cn, H, W, 3)).astype(np.float32))
    y_np = np.clip(x.cpu().numpy() + rng.normal(0, 0.1, (Cn, H, W, 3)), 0, 1).astype(np.float32)
    y = dev(y_np)
    s0, v0 = ops.loss_l1_ssim(ctx, x, y, 0.8, 0.2)
    mom = ops.gt_moments(ctx, y)
    ops.set_gt_moments(ctx, y, mom)
    try:
        s1, v1 = ops.loss_l1_ssim(ctx, x, y, 0.8, 0.2)
        s1b, v1b = ops.loss_l1_ssim(ctx, x[Cn - 1:], y[Cn - 1:], 0.8, 0.2)      # the last view alone: an offset into gt
        other = y.clone()                                                         # another buffer: moments not used, same result
        s2, v2 = ops.loss_l1_ssim(ctx, x, other, 0.8, 0.2)
    finally:
        ops.set_gt_moments(ctx, None, None)
    torch.cuda.synchronize()
    for s_, v_ in ((s1, v1), (s2, v2)):
        assert torch.equal(v_.view(torch.int32), v0.view(torch.int32))
        # (the sums are double-precision atomics over the strips: order-dependent in the last bits only)
        assert torch.allclose(s_, s0, rtol=1e-13, atol=0)
    assert torch.equal(v1b.view(torch.int32), v0[Cn - 1:].view(torch.int32))
    # the moments against a float64 separable convolution
    g = np.exp(-0.5 * ((np.arange(11) - 5) / 1.5) ** 2); g /= g.sum()
    M = mom.cpu().numpy().astype(np.float64)
    yd = y_np.astype(np.float64)

    def conv(a):
        out = np.zeros((Cn, H - 10, W - 10, 3))
        tmp = sum(g[k] * a[:, :, k:k + W - 10] for k in range(11))
        out = sum(g[k] * tmp[:, k:k + H - 10] for k in range(11))
        return out
    assert np.abs(M[:, 5:H - 5, 5:W - 5, :, 0] - conv(yd)).max() < 2e-6
    assert np.abs(M[:, 5:H - 5, 5:W - 5, :, 1] - conv(yd * yd)).max() < 2e-6
    border = np.ones((H, W), bool); border[5:H - 5, 5:W - 5] = False
    assert np.all(M[:, border] == 0.0)


def test_gt_moments_change_no_bit_of_a_training_step():
    """The same through st3r_gs_train_fwd_bwd, in one pass and with the views walked in two chunks (debug flag 32: the second
    chunk's ground-truth pointer is a whole-view offset): gradients and loss bit for bit.  (A context of its own: the chunk
    count sticks to a context.)"""
    from starst3r_amd import ops
    ctx = ops.Context("cuda:0")
    g, w2c, Ks, W, H = make("medium")
    N = g["means"].shape[0]
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, _, _ = ops.render(ctx, P, vm, K, campos, W, H)
    torch.manual_seed(5)
    gt = torch.clamp(rgb + 0.1 * torch.randn_like(rgb), 0, 1).contiguous()
    mom = ops.gt_moments(ctx, gt)
    out = {}
    try:
        for flag in (0, 32):
            for use in (False, True):
                ops.set_debug(ctx, flag)
                ops.set_gt_moments(ctx, gt if use else None, mom if use else None)
                grads = torch.full((23 * N,), float("nan"), device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
                ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
                torch.cuda.synchronize()
                out[(flag, use)] = (grads, float(loss[0]))
    finally:
        ops.set_debug(ctx, 0)
        ops.set_gt_moments(ctx, None, None)
    for flag in (0, 32):
        (g0, l0), (g1, l1) = out[(flag, False)], out[(flag, True)]
        assert bool(torch.isfinite(g1).all())
        assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)), flag
        assert abs(l0 - l1) <= 1e-6 * abs(l0)


def test_adam_vs_oracle_and_torch(ctx):
    from starst3r_amd import ops
    N = 333
    rng = np.random.default_rng(2)
    g0 = synth.make_gaussians(N, seed=9)
    P = {k: dev(v) for k, v in g0.items()}
    ref = {k: v.copy() for k, v in g0.items()}
    m = torch.zeros(23 * N, device="cuda:0"); v = torch.zeros(23 * N, device="cuda:0")
    m_o = np.zeros(23 * N, np.float32); v_o = np.zeros(23 * N, np.float32)
    blocks = [("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1)]
    for step in range(1, 5):
        gr = (rng.standard_normal(23 * N) * 10.0 ** rng.integers(-5, 1, 23 * N)).astype(np.float32)
        ops.adam_step(ctx, P, dev(gr), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        off = 0
        for name, w in blocks:
            sl = slice(off, off + w * N)
            p = ref[name].reshape(-1); mm = m_o[sl]; vv = v_o[sl]
            go.adam(p, gr[sl], mm, vv, 1e-3, 0.9, 0.999, 1e-8, step)
            m_o[sl] = mm; v_o[sl] = vv
            off += w * N
        p = np.ascontiguousarray(ref["shN"][:, :4, :]).reshape(-1); sl = slice(off, off + 12 * N)
        mm = m_o[sl]; vv = v_o[sl]
        go.adam(p, gr[sl], mm, vv, 1e-3, 0.9, 0.999, 1e-8, step)
        m_o[sl] = mm; v_o[sl] = vv
        ref["shN"][:, :4, :] = p.reshape(N, 4, 3)
    torch.cuda.synchronize()
    for name in ("means", "quats", "scales", "opacities", "shN"):
        np.testing.assert_allclose(P[name].cpu().numpy(), ref[name], rtol=0, atol=3e-7, err_msg=name)
    assert np.array_equal(P["shN"].cpu().numpy()[:, 4:], g0["shN"][:, 4:])  # rows 4..23 untouched
    np.testing.assert_allclose(m.cpu().numpy(), m_o, rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(v.cpu().numpy(), v_o, rtol=1e-5, atol=1e-20)


def test_empty_and_culled(ctx):
    """All Gaussians behind the camera: zero intersections, black image, zero gradients."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make("small")
    g["means"][:, :] = g["means"] * 0.01 + np.array([100.0, 0, 0], np.float32)  # far outside every frustum
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks,
                                            W, H)
    P, rgb, alpha, info = run_hip(ctx, g, w2c, Ks, W, H)
    assert meta["isect_ids"].size == info["isect_ids"].numel()
    assert np.array_equal(info["isect_offsets"].cpu().numpy(), meta["isect_offsets"])
    assert float(rgb.abs().max()) == 0.0 or meta["isect_ids"].size > 0


def test_fused_step_on_edge_cases(ctx):
    """The fused path (cell-list forward, two-level sort, exact culling) on inputs the regular scenes do not reach:
    (a) nothing visible -- zero records: black image, finite loss, only the regularisers' gradients;
    (b) an image whose size is no multiple of the 16-pixel tile or of the 4-pixel cell (partial tiles / cells on two
        borders) with more than one batch of 256 records per tile: fused images bit-identical to the staged path's."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make("small")
    N, Cn = g["means"].shape[0], w2c.shape[0]
    far = {k: v.copy() for k, v in g.items()}
    far["means"] = (far["means"] * 0.01 + np.array([100.0, 0, 0], np.float32)).astype(np.float32)
    P = {k: dev(v) for k, v in far.items()}
    vm, K = dev(w2c), dev(Ks)
    gt = torch.rand((Cn, H, W, 3), device="cuda:0")
    grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    st = ops.train_fwd_bwd(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    torch.cuda.synchronize()
    assert st["n_isects"] == 0 and np.isfinite(float(loss))
    assert float(ops.peek(ctx, 8, Cn * H * W * 3, torch.float32).abs().max()) == 0.0
    G = ops.split_grads(grads, N)
    assert float(G["means"].abs().max()) == 0.0 and float(G["sh"].abs().max()) == 0.0
    assert float(G["opacities"].abs().max()) > 0.0          # d/d opacity of the sigmoid regulariser
    # (b)
    from st3r_synth import synth
    W2, H2 = 150, 90
    g2, w2c2, Ks2 = synth.make_scene(6000, 2, W2, H2, seed=8, scale_lo=0.02, scale_hi=0.12)
    P2 = {k: dev(v) for k, v in g2.items()}
    vm2, K2 = dev(w2c2), dev(Ks2)
    rgb, alpha, info = ops.rasterization(ctx, P2["means"], P2["quats"], P2["scales"], P2["opacities"], P2["shN"], vm2, K2, W2, H2)
    per_tile = info["isect_ids"].numel() / (2 * ((W2 + 15) // 16) * ((H2 + 15) // 16))
    assert per_tile > 300                                   # several batches per tile
    gt2 = torch.clamp(rgb + 0.1 * torch.randn_like(rgb), 0, 1).contiguous()
    grads2 = torch.empty(23 * 6000, device="cuda:0")
    ops.train_fwd_bwd(ctx, P2, vm2, K2, ops.camera_positions(vm2), gt2, W2, H2, 0.2, 0.01, 0.01, grads2, loss)
    torch.cuda.synchronize()
    for which, full in ((8, rgb), (9, alpha)):
        got = ops.peek(ctx, which, full.numel(), torch.float32)
        assert torch.equal(got.view(torch.int32), full.reshape(-1).view(torch.int32))
    assert bool(torch.isfinite(grads2).all())


@pytest.mark.parametrize("name", ["small", "medium", "many", "wide"] + FUZZ)
def test_fused_train_gradients_equal_stage_path(ctx, name):
    """The fused train step culls (record, tile) pairs whose alpha >= 1/255 box misses the tile and sorts in
    two levels; its gradients must equal the reference-exact stage path's (the dropped pairs fail the alpha
    test on all 256 pixels).  Also pins the loss against the oracle's composite loss."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    N, Cn = g["means"].shape[0], w2c.shape[0]
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, W, H)
    torch.manual_seed(11)
    gt = torch.clamp(rgb + 0.1 * torch.randn_like(rgb), 0, 1).contiguous()
    sums, v_rgb = ops.loss_l1_ssim(ctx, rgb, gt, 0.8, 0.2)
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                             info["_last_ids"], v_rgb, None, info["_cum_tiles"], Cn, W, H)
    ref = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, campos, W, H,
                             info["_splats"], v_splats, float(Cn), 0.01, 0.01)
    grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    st = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    torch.cuda.synchronize()
    assert st["n_isects_ref"] == info["isect_ids"].numel() and st["n_isects"] <= st["n_isects_ref"]
    if name == "wide":
        assert st["n_isects"] > 6 * st["n_visible"]       # (the slot gather's wide form: more than six slots per pair)
    # the culled render is the un-culled render, bit for bit ...
    for which, full in ((8, rgb), (9, alpha)):
        got = ops.peek(ctx, which, full.numel(), torch.float32)
        assert torch.equal(got.view(torch.int32), full.reshape(-1).view(torch.int32))
    # ... so the gradients differ only through the grouping of the float sums (other batches of 32 records); the
    # stress scenes' needle-shaped Gaussians sum terms far above the result and get a wider bound
    scale = float(ref.abs().max())
    assert float((grads - ref).abs().max()) <= (2e-3 if name.startswith("fuzz") else 2e-5) * scale
    s = sums.cpu().numpy()
    expect = sum(0.8 * s[c, 0] / (H * W * 3) + 0.2 * (1 - s[c, 1] / ((H - 10) * (W - 10) * 3)) for c in range(Cn))
    expect += Cn * (0.01 * float(torch.sigmoid(P["opacities"]).mean()) + 0.01 * float(torch.exp(P["scales"]).mean()))
    assert abs(float(loss[0]) - expect) <= 1e-5 * abs(expect)


@pytest.mark.parametrize("name", ["small", "medium", "many", "wide"] + FUZZ)
def test_cell_list_forward_equals_quadrant_forward(ctx, name):
    """The fused training path blends with 4x4-cell lists (gs_blend_cells.hip: four records per trip, one per 16-lane row,
    exec-masked tests); st3r_gs_render takes the same kernel under debug flag 512.  Same records per pixel in the same
    order with the same arithmetic: the images are bit-identical to the quadrant kernel's, with the cell test on (the
    exact row-strip test must be a superset of the hit cells) and off (flag 1: every record in every list)."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    r0, a0, _ = ops.render(ctx, P, vm, K, campos, W, H)
    for flags in (512, 513):
        ops.set_debug(ctx, flags)
        try:
            r1, a1, _ = ops.render(ctx, P, vm, K, campos, W, H)
        finally:
            ops.set_debug(ctx, 0)
        assert torch.equal(r1.view(torch.int32), r0.view(torch.int32)), flags
        assert torch.equal(a1.view(torch.int32), a0.view(torch.int32)), flags


@pytest.mark.parametrize("name", ["small", "ragged", "medium", "one", "wide"] + FUZZ[:6])
def test_segmented_level1_sort_changes_no_bit(ctx, name):
    """Round 6: the training calls sort the pairs of every camera as a segment of its own, on depth codes biased by the
    smallest one of the call, in as many 8-bit passes as that range needs (three at SYNTH-1M instead of four; decided on the
    device).  Debug flag 4 keeps the (camera | depth) keys in four passes: same permutation, so the sorted record list, the
    images, the gradients and the loss are the same bits -- on narrow depth ranges (1 - 3 passes) and wide ones (4)."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make(name)
    N, Cn = g["means"].shape[0], w2c.shape[0]
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, _, _ = ops.render(ctx, P, vm, K, campos, W, H)
    torch.manual_seed(5)
    gt = torch.clamp(rgb + 0.1 * torch.randn_like(rgb), 0, 1).contiguous()
    out = []
    try:
        for flag in (4, 0):
            ops.set_debug(ctx, flag)
            grads = torch.full((23 * N,), float("nan"), device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
            st = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
            torch.cuda.synchronize()
            flat = ops.peek(ctx, 0, st["n_isects"]).clone()
            krange = ops.peek(ctx, 10, 16)[8:11].tolist()
            out.append((grads, float(loss[0]), flat, st, krange))
    finally:
        ops.set_debug(ctx, 0)
    (g0, l0, f0, st0, _), (g1, l1, f1, st1, kr) = out
    assert st0 == st1 and l0 == l1
    assert torch.equal(f0, f1)
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32))
    assert 1 <= kr[2] <= 4 and (kr[1] < (1 << (8 * kr[2]))) and (kr[2] == 1 or kr[1] >= (1 << (8 * (kr[2] - 1))))


def test_train_step_end_to_end(ctx):
    """Fused fwd+bwd+Adam: the first-iteration loss equals the oracle's composite loss and the
    loss goes down over 30 iterations (starster/gs.py:143-161 semantics)."""
    from starst3r_amd import ops
    N, V, W, H = 3000, 3, 128, 96
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=4, scale_lo=0.01, scale_hi=0.05)
    gt_g = synth.perturb_for_gt(g, sigma=0.01)
    gt_img, _, _ = go.rasterization(gt_g["means"], gt_g["quats"], gt_g["scales"], gt_g["opacities"], gt_g["shN"], w2c,
                                    Ks, W, H)
    gt_img = np.clip(gt_img, 0, 1)
    # oracle loss for iteration 0
    rgb_o, _, _ = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
    loss_o = 0.0
    for c in range(V):
        l1, ss, _ = go.l1_ssim(rgb_o[c], gt_img[c], want_grad=False)
        loss_o += 0.8 * l1 + 0.2 * (1 - ss)
        loss_o += 0.01 * np.mean(1 / (1 + np.exp(-g["opacities"].astype(np.float64))))
        loss_o += 0.01 * np.mean(np.exp(g["scales"].astype(np.float64)))
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    gt = dev(gt_img)
    grads = torch.empty(23 * N, device="cuda:0"); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    losses = torch.zeros(30, device="cuda:0")
    for it in range(30):
        ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, losses[it:it + 1])
        ops.adam_step(ctx, P, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1)
    torch.cuda.synchronize()
    L = losses.cpu().numpy()
    assert abs(L[0] - loss_o) / loss_o < 1e-4, (L[0], loss_o)
    assert L[-1] < L[0]
    assert np.all(np.isfinite(L))


def test_train_gradients_are_bit_reproducible(ctx):
    """Two fused forward+backward passes over the same inputs give the same gradient bits (no float atomics on the path:
    per-wave accumulator rows, stamped slots summed in order, fixed-order regulariser sums)."""
    from starst3r_amd import ops
    g, w2c, Ks, W, H = make("many")
    N = g["means"].shape[0]
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, _, _ = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, W, H)
    gt = torch.clamp(rgb + 0.1 * torch.randn(rgb.shape, device=rgb.device,
                                             generator=torch.Generator(device=rgb.device).manual_seed(9)), 0, 1).contiguous()
    out = []
    for _ in range(2):
        grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
        ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
        out.append(grads)
    torch.cuda.synchronize()
    assert torch.equal(out[0].view(torch.int32), out[1].view(torch.int32))


@pytest.mark.parametrize("case", ["wide", "flag64", "views16"])
def test_fused_front_end_variants_equal_the_stage_path(ctx, case):
    """The three shapes of the round-4 front end that SYNTH-1M does not reach: (wide) a tile grid of more than 255 tiles
    a side -- 64-bit packed rectangles instead of 32-bit --; (flag64) the same 64-bit form forced on a small image together
    with the backward's separate rectangle / slot-base gathers; (views16) sixteen views -- 64-bit level-1 keys and two
    cameras per XCD queue of the emission kernels.  Fused gradients against the reference-exact stage path."""
    from starst3r_amd import ops
    from st3r_synth import synth
    if case == "wide":
        N, V, W, H = 3000, 2, 4112, 48          # 257 x 3 tiles
    elif case == "views16":
        N, V, W, H = 1500, 16, 96, 64
    else:
        N, V, W, H = 3000, 2, 160, 96
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=21, scale_lo=0.01, scale_hi=0.06)
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, W, H)
    assert info["isect_ids"].numel() > 1000
    torch.manual_seed(3)
    gt = torch.clamp(rgb + 0.1 * torch.randn_like(rgb), 0, 1).contiguous()
    sums, v_rgb = ops.loss_l1_ssim(ctx, rgb, gt, 0.8, 0.2)
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                             info["_last_ids"], v_rgb, None, info["_cum_tiles"], V, W, H)
    ref = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, campos, W, H,
                             info["_splats"], v_splats, float(V), 0.01, 0.01)
    grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    if case == "flag64":
        ops.set_debug(ctx, 64)
    try:
        st = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
        torch.cuda.synchronize()
    finally:
        ops.set_debug(ctx, 0)
    assert st["n_isects_ref"] == info["isect_ids"].numel() and 0 < st["n_isects"] <= st["n_isects_ref"]
    for which, full in ((8, rgb), (9, alpha)):
        got = ops.peek(ctx, which, full.numel(), torch.float32)
        assert torch.equal(got.view(torch.int32), full.reshape(-1).view(torch.int32))
    assert float((grads - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def _fuzz_shapes():
    rng = np.random.default_rng(20260929)
    shapes = []
    for _ in range(24):
        V = int(rng.integers(1, 12))                      # <= 8 views: 32-bit level-1 keys; more: 64-bit
        W = int(rng.integers(17, 420)); H = int(rng.integers(17, 300))    # ragged: not multiples of the 16-pixel tile
        N = int(rng.choice([1, 7, 63, 64, 65, 257, 1000, 4095, 4097, 9000]))
        lo = float(rng.choice([0.002, 0.01, 0.05])); hi = lo * float(rng.choice([1.5, 4.0, 10.0]))
        shapes.append((N, V, W, H, lo, hi, int(rng.integers(0, 1000))))
    return shapes


@pytest.mark.parametrize("N,V,W,H,lo,hi,seed", _fuzz_shapes())
def test_fused_front_end_fuzz_against_the_stage_path(ctx, N, V, W, H, lo, hi, seed):
    """Random ragged shapes through the fused training call (single-pass scans, gather + owner-scan emission, packed
    rectangles, two-level sort, cell-list forward) against the reference-exact stage path: same images bit for bit, the
    same intersection counts, gradients to 2e-5 of the scale.  Covers one Gaussian, fewer Gaussians than a wave, scenes
    whose Gaussians are all culled in some view, images smaller than two tiles, 1 to 11 views."""
    from starst3r_amd import ops
    from st3r_synth import synth
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=seed, scale_lo=lo, scale_hi=hi)
    P = {k: dev(v) for k, v in g.items()}
    vm, K = dev(w2c), dev(Ks)
    campos = ops.camera_positions(vm)
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, W, H)
    gen = torch.Generator(device=rgb.device).manual_seed(seed)
    gt = torch.clamp(rgb + 0.1 * torch.randn(rgb.shape, device=rgb.device, generator=gen), 0, 1).contiguous()
    sums, v_rgb = ops.loss_l1_ssim(ctx, rgb, gt, 0.8, 0.2)
    n_ref = info["isect_ids"].numel()
    if n_ref:
        v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                                 info["_last_ids"], v_rgb, None, info["_cum_tiles"], V, W, H)
    else:
        v_splats = torch.zeros(N * V * 12, device="cuda:0")
    ref = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], vm, K, campos, W, H,
                             info["_splats"], v_splats, float(V), 0.01, 0.01)
    grads = torch.full((23 * N,), float("nan"), device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
    st = ops.train_fwd_bwd(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    torch.cuda.synchronize()
    assert st["n_isects_ref"] == n_ref and 0 <= st["n_isects"] <= n_ref
    for which, full in ((8, rgb), (9, alpha)):
        got = ops.peek(ctx, which, full.numel(), torch.float32)
        assert torch.equal(got.view(torch.int32), full.reshape(-1).view(torch.int32)), which
    assert bool(torch.isfinite(grads).all())
    assert float((grads - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-30)


def test_scan_status_generation_wraps_cleanly(ctx):
    """The single-pass scan never clears its status words between launches: they carry a 14-bit launch generation, and
    the host clears the block when the generation is about to repeat.  More launches than generations (a training run
    of > 16 000 steps gets there), checked against numpy on both sides of the wrap, with sizes that make the tile count
    -- and so the set of stale words -- vary."""
    from starst3r_amd import ops
    rng = np.random.default_rng(0)
    sizes = [5, 4096, 4097, 30000, 123457]
    data = {n: rng.integers(0, 7, n).astype(np.int32) for n in sizes}
    dv = {n: torch.tensor(a, device="cuda:0") for n, a in data.items()}
    want = {n: np.cumsum(a, dtype=np.int64) for n, a in data.items()}
    for it in range(16500):
        n = sizes[it % len(sizes)]
        check_now = it < 10 or it % 997 == 0 or 16370 <= it <= 16400
        cum, total = ops.isect_scan(ctx, dv[n])
        if check_now:
            assert total == int(want[n][-1]), (it, n)
            assert np.array_equal(cum.cpu().numpy().astype(np.int64), want[n]), (it, n)
