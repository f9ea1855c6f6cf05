"""Oracle-vs-HIP parity at BASELINE sizes (reference: starster/gs.py:76-87, :126-136, :143-161).

  * one view of SYNTH-1M (BASELINE.json configs[2]: 1 M Gaussians, 1920x1080) through oracle/gs_oracle.c -- about 16 s on
    one host core -- against the HIP stage path: packed ids / radii / tile counts / sorted keys / flatten ids / offsets
    bit-exact, pixels within 1e-4 relative on the pixels float32 decides (the oracle reports a per-pixel decision margin);
  * PSNR parity of a 20 000-Gaussian, 4-view, 320x240 training run (configs[1]-like density): 50 iterations of the C
    oracle's forward + backward + torch.optim.Adam on the host against the fused HIP step -- |dPSNR| <= 0.1 dB per view,
    losses within 1 %.  (tests/test_gpu_psnr.py does the same against fp64 autograd at a few hundred Gaussians.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as go
from st3r_synth import synth

DEV = "cuda:0"


def _dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


@pytest.fixture(scope="module")
def synth_1m_view():
    """One of the eight views of SYNTH-1M through the C oracle (forward: ~16 s on one host core), shared by the tests."""
    N, W, H = 1_000_000, 1920, 1080
    g, w2c, Ks = synth.make_scene(N, 8, W, H)
    w2c, Ks = w2c[3:4].copy(), Ks[3:4].copy()
    rgb_o, alpha_o, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H,
                                            want_margin=True)
    return g, w2c, Ks, rgb_o, alpha_o, meta


def test_one_view_of_synth_1m_against_the_c_oracle(synth_1m_view):
    from starst3r_amd import ops
    N, W, H = 1_000_000, 1920, 1080
    g, w2c, Ks, rgb_o, alpha_o, meta = synth_1m_view
    ctx = ops.get_context(DEV)
    P = {k: _dev(v) for k, v in g.items()}
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], _dev(w2c),
                                         _dev(Ks), W, H)
    torch.cuda.synchronize()
    for key in ("camera_ids", "gaussian_ids", "radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert np.array_equal(info[key].cpu().numpy().reshape(-1), np.asarray(meta[key]).reshape(-1)), key
    n_isects = meta["isect_ids"].size
    assert n_isects > 3_000_000
    ok = meta["margin"] > 1e-4
    assert ok.mean() > 0.995
    a, b = rgb.cpu().numpy()[ok], rgb_o[ok]
    assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    assert np.abs(alpha.cpu().numpy()[ok] - alpha_o[ok]).max() <= 1e-4
    # the pixels float32 does not decide stay close as well
    assert np.abs(rgb.cpu().numpy()[~ok] - rgb_o[~ok]).max() <= 5e-3


def test_training_record_list_of_one_view_of_synth_1m_is_the_oracle_list_minus_dead_pairs(synth_1m_view):
    """The fused TRAINING path's record list at full size (the path the headline number times) against the C oracle's sorted
    list: equal, order included, once the pairs the exact culling drops are removed; every dropped pair (a 300 000 sample of
    them) fails the alpha test on all 256 pixels of its tile; tile offsets follow (tests/test_gpu_gs.py:
    _kept_list_against_oracle)."""
    from starst3r_amd import ops
    from test_gpu_gs import _kept_list_against_oracle
    N, W, H = 1_000_000, 1920, 1080
    g, w2c, Ks, rgb_o, alpha_o, meta = synth_1m_view
    ctx = ops.Context(DEV)
    P = {k: _dev(v) for k, v in g.items()}
    vm, K = _dev(w2c), _dev(Ks)
    gt = _dev(np.clip(rgb_o, 0, 1))
    grads = torch.empty(23 * N, device=DEV); loss = torch.zeros(1, device=DEV)
    st = ops.train_fwd_bwd(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
    torch.cuda.synchronize()
    assert st["n_isects_ref"] == meta["isect_ids"].size
    n_ref, n_drop = _kept_list_against_oracle(ctx, g, w2c, Ks, W, H, meta, st["n_isects"], max_dropped=300_000)
    assert n_ref - n_drop == st["n_isects"] and 0.2 * n_ref < n_drop < 0.5 * n_ref   # (SYNTH-1M: 35 % of the pairs are dead)


def test_backward_of_one_view_of_synth_1m_against_the_c_oracle(synth_1m_view):
    """VERDICT r3 'What's weak' 1(i): backward parity against the C oracle stopped at 20 000 Gaussians.  Here: the same
    full-size view (1 M Gaussians, 1920x1080, ~3.4 M intersections, every covered pixel ~36 Gaussians deep) through
    gso_rasterization_backward on one host core (about a minute) against st3r_gs_blend_bwd + st3r_gs_project_sh_bwd:
    per-pair and per-parameter gradients within the bounds of tests/test_gpu_gs.py::test_backward_vs_oracle, error
    DISTRIBUTIONS pinned (median / 99th percentile of the element-wise relative error above 1e-4 of the maximum)."""
    from starst3r_amd import ops
    N, W, H = 1_000_000, 1920, 1080
    g, w2c, Ks, rgb_o, alpha_o, meta = synth_1m_view
    rng = np.random.default_rng(3)
    v_rgb = rng.standard_normal(rgb_o.shape).astype(np.float32)
    v_rgb[~(meta["margin"] > 1e-4)] = 0.0     # no gradient through pixels float32 does not decide (test_backward_vs_oracle)
    ref = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H, meta,
                                    alpha_o, v_rgb, None)
    ctx = ops.get_context(DEV)
    P = {k: _dev(v) for k, v in g.items()}
    rgb, alpha, info = ops.rasterization(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], _dev(w2c),
                                         _dev(Ks), W, H)
    v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                             info["_last_ids"], _dev(v_rgb), None, info["_cum_tiles"], 1, W, H)
    grads = ops.project_sh_bwd(ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], _dev(w2c), _dev(Ks),
                               info["_campos"], W, H, info["_splats"], v_splats)
    torch.cuda.synchronize()
    pid = info["camera_ids"].long() * N + info["gaussian_ids"].long()
    vs = v_splats[pid].cpu().numpy()
    pk = ref["packed"]
    dist = {}

    def close(a, b, name, tol, med_bound, p99_bound):
        scale = np.abs(b).max() + 1e-20
        err = np.abs(a - b).max() / scale
        big = np.abs(b) > 1e-4 * scale
        rel = np.abs(a - b)[big] / np.abs(b)[big]
        med, p99 = float(np.median(rel)), float(np.percentile(rel, 99))
        dist[name] = (float(err), med, p99, int(big.sum()))
        assert err < tol and med <= med_bound and p99 <= p99_bound, (name, err, med, p99)
    # 2e-4 of the tensor's maximum for pairs AND parameters (measured at this size: <= 5.6e-5; round 3's parameter bound was
    # 1e-3), median 2e-6 and p99 1e-4 of the element-wise relative error as on the regular small scenes
    close(vs[:, 0:2], pk["v_means2d"], "v_means2d", 2e-4, 2e-6, 1e-4)
    close(vs[:, 2], pk["v_opacities"], "v_opacities", 2e-4, 2e-6, 1e-4)
    close(vs[:, 3:6], pk["v_conics"], "v_conics", 2e-4, 2e-6, 1e-4)
    close(vs[:, 6:9], pk["v_colors"], "v_colors", 2e-4, 2e-6, 1e-4)
    G = {k: v.cpu().numpy() for k, v in ops.split_grads(grads, N).items()}
    for k in ("means", "quats", "scales", "opacities", "sh"):
        close(G[k], ref[k], k, 2e-4, 2e-6, 1e-4)
    print("SYNTH-1M view, gradient error (max / tensor max, median rel, p99 rel, elements):",
          {k: (f"{a:.1e}", f"{b:.1e}", f"{c:.1e}", n) for k, (a, b, c, n) in dist.items()})


def _psnr(a, b):
    return 10.0 * np.log10(1.0 / max(float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)), 1e-20))


def test_psnr_parity_against_the_c_oracle_train_loop():
    from starst3r_amd import ops
    N, V, W, H, iters = 20000, 4, 320, 240, 50
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=7, scale_lo=0.004, scale_hi=0.03)
    gt_g = synth.perturb_for_gt(g, sigma=0.004)
    gt, _, _ = go.rasterization(gt_g["means"], gt_g["quats"], gt_g["scales"], gt_g["opacities"], gt_g["shN"], w2c, Ks, W, H)
    gt = np.clip(gt, 0, 1).astype(np.float32)
    keys = ("means", "quats", "scales", "opacities", "shN")
    # ---- host: C oracle forward + loss + backward, one torch.optim.Adam per tensor (starster/gs.py:37,143-161)
    Pt = {k: torch.tensor(g[k], dtype=torch.float32, requires_grad=True) for k in keys}
    opts = [torch.optim.Adam([Pt[k]], lr=1e-3) for k in keys]
    loss_ref = []
    for _ in range(iters):
        cur = {k: Pt[k].detach().numpy() for k in keys}
        rgb, alpha, meta = go.rasterization(cur["means"], cur["quats"], cur["scales"], cur["opacities"], cur["shN"], w2c, Ks,
                                            W, H)
        v_rgb = np.zeros_like(rgb); loss = 0.0
        for c in range(V):
            l1, ss, vr = go.l1_ssim(rgb[c], gt[c], 0.8, 0.2)
            loss += 0.8 * l1 + 0.2 * (1 - ss); v_rgb[c] = vr
        G = go.rasterization_backward(cur["means"], cur["quats"], cur["scales"], cur["opacities"], cur["shN"], w2c, Ks, W,
                                      H, meta, alpha, v_rgb, None)
        sg = 1 / (1 + np.exp(-cur["opacities"].astype(np.float64))); ex = np.exp(cur["scales"].astype(np.float64))
        loss += V * (0.01 * sg.mean() + 0.01 * ex.mean())            # both regularisers once per view (gs.py:150-152)
        G["opacities"] = G["opacities"] + V * 0.01 * sg * (1 - sg) / N
        G["scales"] = G["scales"] + V * 0.01 * ex / (3 * N)
        sh_grad = np.zeros_like(cur["shN"]); sh_grad[:, :4] = np.asarray(G["sh"]).reshape(N, 4, 3)
        for k, gk in (("means", G["means"]), ("quats", G["quats"]), ("scales", G["scales"]), ("opacities", G["opacities"]),
                      ("shN", sh_grad)):
            Pt[k].grad = torch.tensor(np.asarray(gk, np.float32).reshape(Pt[k].shape))
        for o in opts:
            o.step()
        loss_ref.append(float(loss))
    ref = {k: Pt[k].detach().numpy() for k in keys}
    # ---- device: the fused step
    ctx = ops.get_context(DEV)
    P = {k: _dev(v) for k, v in g.items()}
    vm, K, GT = _dev(w2c), _dev(Ks), _dev(gt)
    campos = ops.camera_positions(vm)
    grads = torch.empty(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    losses = torch.zeros(iters, device=DEV)
    for it in range(iters):
        ops.train_step(ctx, P, vm, K, campos, GT, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                       losses[it:it + 1])
    torch.cuda.synchronize()
    hip = {k: t.cpu().numpy() for k, t in P.items()}
    loss_hip = losses.cpu().numpy()

    def render(Q):
        return go.rasterization(Q["means"], Q["quats"], Q["scales"], Q["opacities"], Q["shN"], w2c, Ks, W, H)[0]
    r0, r_ref, r_hip = render(g), render(ref), render(hip)
    rows = []
    for c in range(V):
        p0, pr, ph = _psnr(r0[c], gt[c]), _psnr(r_ref[c], gt[c]), _psnr(r_hip[c], gt[c])
        rows.append((c, round(p0, 3), round(pr, 3), round(ph, 3)))
        assert pr > p0 + 0.3, rows
        assert abs(ph - pr) <= 0.1, rows
    print("view, PSNR before / C oracle + torch Adam / HIP:", rows)
    assert abs(loss_hip[0] - loss_ref[0]) <= 1e-4 * abs(loss_ref[0])
    np.testing.assert_allclose(loss_hip, loss_ref, rtol=1e-2)
