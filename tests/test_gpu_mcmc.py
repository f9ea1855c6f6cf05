"""GPU parity of the MCMC refinement hooks (csrc/mcmc.hip through the C-ABI) against oracle/mcmc_oracle.py:
integer sampling state bit-exact, relocated parameters within 2e-5, plus the Scene-level behaviour of
run_3dgs_optim(enable_pruning=True) (growth, optimiser-state bookkeeping, replica determinism)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mcmc_oracle as mo

DEV = "cuda:0"


def make_params(N, n_dead, seed):
    rng = np.random.default_rng(seed)
    P = {"means": rng.normal(size=(N, 3)), "quats": rng.normal(size=(N, 4)), "scales": rng.normal(-4, 0.3, (N, 3)),
         "opacities": np.clip(rng.normal(0, 2, N), -4, 4), "sh0": rng.normal(size=(N, 1, 3)),
         "shN": rng.normal(size=(N, 24, 3))}
    P = {k: v.astype(np.float32) for k, v in P.items()}
    dead = rng.choice(N, n_dead, replace=False)
    P["opacities"][dead] = rng.uniform(-9, -5.5, n_dead).astype(np.float32)
    return P


def to_dev(P):
    return {k: torch.from_numpy(v.copy()).to(DEV) for k, v in P.items()}


def adam_blocks(N, m):
    """[23N] block buffer -> dict of per-parameter views (numpy)."""
    m = m.cpu().numpy()
    out, off = {}, 0
    for k, w in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1), ("shN", 12)):
        out[k] = m[off * N:(off + w) * N].reshape(N, w); off += w
    return out


@pytest.mark.parametrize("N,n_dead", [(5000, 300), (257, 1), (4096, 0), (100000, 9000)])
def test_relocate_matches_oracle(N, n_dead):
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    P = make_params(N, n_dead, N)
    D = to_dev(P)
    m = torch.ones(23 * N, device=DEV); v = torch.full((23 * N,), 2.0, device=DEV)
    n = ops.mcmc_relocate(ctx, D, m, v, 0.005, seed=0xABCDEF0123, step=17)
    assert n == n_dead
    # integer state: weights within one quantum of the oracle's, everything derived from them bit-exact
    cum = ops.peek(ctx, 4, N, torch.int64).cpu().numpy().astype(np.uint64)
    w_dev = np.diff(np.concatenate([[0], cum.astype(np.int64)]))
    w_ref, dead_ref = mo.weights(P["opacities"], 0.005, True)
    assert np.abs(w_dev - w_ref.astype(np.int64)).max() <= 2
    dead_dev = ops.peek(ctx, 5, N, torch.int32).cpu().numpy().astype(bool)
    np.testing.assert_array_equal(dead_dev, dead_ref)
    Q = {k: v_.copy() for k, v_ in P.items()}
    adam = {k: (np.ones_like(x), np.full_like(x, 2.0)) for k, x in Q.items()}
    dead_ids, sampled = mo.relocate(Q, adam, 0.005, seed=0xABCDEF0123, step=17, cum_override=cum)
    if n_dead:
        st = ops.peek(ctx, 6, 2 * N, torch.int32).cpu().numpy()
        np.testing.assert_array_equal(st[:n_dead], sampled)
        np.testing.assert_array_equal(st[N:N + n_dead], dead_ids)
        counts = ops.peek(ctx, 7, N, torch.int32).cpu().numpy()
        np.testing.assert_array_equal(counts, np.bincount(sampled, minlength=N))
    for k in Q:
        np.testing.assert_allclose(D[k].cpu().numpy(), Q[k], rtol=2e-5, atol=2e-5, err_msg=k)
    # dead rows are exact copies of their (updated) sources; untouched rows are untouched
    if n_dead:
        for k in Q:
            got = D[k].cpu().numpy()
            np.testing.assert_array_equal(got[dead_ids], got[sampled])
            keep = np.setdiff1d(np.arange(N), np.concatenate([dead_ids, sampled]))
            np.testing.assert_array_equal(got[keep], P[k][keep])
    # Adam moments: sources zeroed, everything else (dead rows included) kept
    mb, vb = adam_blocks(N, m), adam_blocks(N, v)
    for k in ("means", "quats", "scales", "opacities"):
        np.testing.assert_array_equal(mb[k], adam[k][0].reshape(N, -1))
        np.testing.assert_array_equal(vb[k], adam[k][1].reshape(N, -1))
    np.testing.assert_array_equal(mb["shN"], adam["shN"][0].reshape(N, 72)[:, :12])


def test_relocate_with_nothing_alive_is_a_no_op():
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    P = make_params(300, 0, 3); P["opacities"][:] = -9
    D = to_dev(P)
    assert ops.mcmc_relocate(ctx, D, None, None, 0.005, 1, 1) == 300
    for k in P:
        np.testing.assert_array_equal(D[k].cpu().numpy(), P[k])


@pytest.mark.parametrize("N,n_new", [(4000, 200), (50000, 2500), (64, 3)])
def test_add_matches_oracle(N, n_new):
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    P = make_params(N, N // 50, N + 1)
    D = {k: torch.cat([torch.from_numpy(v), torch.full((n_new,) + v.shape[1:], float("nan"))]).to(DEV)
         for k, v in P.items()}
    ops.mcmc_add(ctx, D, N, n_new, 0.005, seed=77, step=5)
    cum = ops.peek(ctx, 4, N, torch.int64).cpu().numpy().astype(np.uint64)
    Q = {k: v.copy() for k, v in P.items()}
    grown, sampled = mo.add_new(Q, n_new, 0.005, seed=77, step=5, cum_override=cum)
    np.testing.assert_array_equal(ops.peek(ctx, 6, n_new, torch.int32).cpu().numpy(), sampled)
    for k in grown:
        got = D[k].cpu().numpy()
        np.testing.assert_allclose(got, grown[k], rtol=2e-5, atol=2e-5, err_msg=k)
        np.testing.assert_array_equal(got[N:], got[sampled])


def test_sampling_distribution_on_device():
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    N, n_new = 32, 60000
    P = make_params(N, 0, 9)
    D = {k: torch.cat([torch.from_numpy(v), torch.zeros((n_new,) + v.shape[1:])]).to(DEV) for k, v in P.items()}
    ops.mcmc_add(ctx, D, N, n_new, 0.005, seed=1, step=0)
    counts = ops.peek(ctx, 7, N, torch.int32).cpu().numpy()
    p = mo.sigmoid32(P["opacities"]).astype(np.float64); p /= p.sum()
    chi2 = ((counts - n_new * p) ** 2 / (n_new * p)).sum()
    assert counts.sum() == n_new and chi2 < 75  # 31 dof: mean 31, sd 7.9


def test_noise_matches_oracle():
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    N = 20000
    P = make_params(N, 2000, 4)   # the gate is ~0 for opaque Gaussians, ~0.5..1 for the dead ones
    D = to_dev(P)
    ops.mcmc_noise(ctx, D, 1e-3 * 5e5, seed=31337, step=9)
    delta = D["means"].cpu().numpy().astype(np.float64) - P["means"]
    want = mo.noise_delta(P["quats"], P["scales"], P["opacities"], 1e-3 * 5e5, 9, 31337)
    scale = np.abs(want).max()
    assert scale > 1e-4
    # the increment is added to float32 means of magnitude ~1: compare with the float32 resolution of the means
    np.testing.assert_allclose(delta, want, rtol=1e-3, atol=4e-7)
    moved = np.abs(want).max(1) > 1e-5
    assert moved.sum() > 1000
    np.testing.assert_allclose(delta[moved], want[moved], rtol=2e-2, atol=4e-7)
    for k in ("quats", "scales", "opacities"):
        np.testing.assert_array_equal(D[k].cpu().numpy(), P[k])


def _tiny_scene():
    import starst3r_amd as st
    from st3r_synth.synth_model import SyntheticPairwiseModel
    model = SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2)
    sc = st.Scene(device=DEV)
    sc.add_images(model, [torch.zeros(3, 96, 128) for _ in range(2)])
    sc.init_3dgs()
    return sc


def test_run_3dgs_optim_with_pruning_grows_and_stays_consistent():
    sc = _tiny_scene()
    N0 = sc.gaussians["means"].shape[0]
    sc.strategy.refine_start_iter, sc.strategy.refine_every, sc.strategy.cap_max = 2, 3, int(1.08 * N0)
    with torch.no_grad():
        sc.gaussians["opacities"].data[:50] = -7.0   # dead for the strategy (and invisible for the renderer)
    sh_tail = sc.gaussians["shN"].data[:, 4:].clone()
    losses = sc.run_3dgs_optim(8, enable_pruning=True)   # refine at steps 3 and 6
    assert len(losses) == 8 and all(np.isfinite(losses))
    N1 = sc.gaussians["means"].shape[0]
    assert N1 == int(1.08 * N0)                          # +5 %, then capped
    st = sc._gs_optim
    assert st.N == N1 and st.m.numel() == 23 * N1 and st.grads.numel() == 23 * N1
    for k in ("means", "scales", "quats", "opacities", "sh0", "shN"):
        assert sc.gaussians[k].shape[0] == N1 and sc.gaussians[k].requires_grad
        assert sc.optimizers[k].param_groups[0]["params"][0] is sc.gaussians[k]
    assert sc.strategy_state["n_added"] == N1 - int(1.05 * N0)
    assert (torch.sigmoid(sc.gaussians["opacities"].data) > 0.005).all()   # the dead ones were relocated
    # unused SH rows ride along unchanged (the 50 relocated rows took their source's)
    assert torch.equal(sc.gaussians["shN"].data[50:N0, 4:], sh_tail[50:])
    img, alpha, _ = sc.render_3dgs_original(128, 96)
    assert torch.isfinite(img).all()
    # default window: nothing but noise before step 500 (B-6: the step restarts per call)
    sc.strategy.refine_start_iter, sc.strategy.refine_every = 500, 100
    sc.run_3dgs_optim(3, enable_pruning=True)
    assert sc.gaussians["means"].shape[0] == N1


def test_refinement_is_deterministic_across_replicas():
    """Same parameters + same seed/step -> identical decisions and values (what view-sharded ranks rely on)."""
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    N, n_new = 30000, 1500
    P = make_params(N, 2500, 8)
    outs = []
    for _ in range(2):
        D = {k: torch.cat([torch.from_numpy(v), torch.zeros((n_new,) + v.shape[1:])]).to(DEV) for k, v in P.items()}
        m = torch.ones(23 * N, device=DEV); v = torch.ones(23 * N, device=DEV)
        head = {k: t[:N] for k, t in D.items()}
        ops.mcmc_relocate(ctx, head, m, v, 0.005, 5, 2)
        ops.mcmc_add(ctx, D, N, n_new, 0.005, 5, 2)
        ops.mcmc_noise(ctx, D, 500.0, 5, 2)
        outs.append((D, m, v))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
