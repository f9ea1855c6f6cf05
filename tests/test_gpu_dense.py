"""GPU parity of the dense point extraction (st3r_dense_unproject / st3r_dense_clean) against
oracle/dense_oracle.py, and the Scene-level seeding path built on it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dense_oracle as do

DEV = "cuda:0"


def look_at(eye):
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.eye(4); m[:3, 0] = right; m[:3, 1] = down; m[:3, 2] = fwd; m[:3, 3] = eye
    return m


def make_problem(C=4, H=40, W=56, G=35, seed=0):
    rng = np.random.default_rng(seed)
    f = 0.9 * W
    cam2w = [look_at(np.array([3 * np.cos(a), 3 * np.sin(a), 0.4 + 0.2 * k])) for k, a in enumerate(np.linspace(0, 1.6, C))]
    Ks = [np.array([[f * (1 + 0.02 * k), 0, W / 2 + k], [0, f * (1 + 0.02 * k), H / 2 - k], [0, 0, 1.0]]) for k in range(C)]
    ys, xs = np.mgrid[0:H, 0:W]
    pix = np.stack([xs.reshape(-1) + 0.5, ys.reshape(-1) + 0.5], -1).astype(np.float32)
    core = rng.uniform(0.8, 1.2, (C, G)).astype(np.float32)
    A = rng.uniform(-0.1, 0.1, C).astype(np.float32); B = rng.uniform(2.0, 3.5, C).astype(np.float32)
    idxs = [rng.integers(0, G, H * W).astype(np.int32) for _ in range(C)]
    # smooth-ish surfaces with fliers in front of them (the points clean_pointcloud is there to catch)
    offs = [(1 + 0.05 * np.sin(xs / 7 + k) * np.cos(ys / 5)).reshape(-1).astype(np.float32) for k in range(C)]
    for k in range(C):
        fl = rng.choice(H * W, H * W // 15, replace=False)
        offs[k][fl] *= rng.uniform(0.5, 0.8, len(fl)).astype(np.float32)
    confs = [rng.uniform(1.0, 4.0, H * W).astype(np.float32) for _ in range(C)]
    base_f = rng.uniform(0.8, 1.1, C).astype(np.float32) * f
    cam = np.zeros((C, 24), np.float32)
    for k in range(C):
        cam[k, :9] = cam2w[k][:3, :3].reshape(-1); cam[k, 9:12] = cam2w[k][:3, 3]
        cam[k, 12] = Ks[k][0, 0]; cam[k, 13] = Ks[k][0, 2]; cam[k, 14] = Ks[k][1, 2]; cam[k, 15] = A[k]; cam[k, 16] = B[k]
    return dict(C=C, H=H, W=W, G=G, pix=pix, core=core, idxs=idxs, offs=offs, confs=confs, base_f=base_f, cam=cam,
                Ks=Ks, cam2w=cam2w, A=A, B=B)


@pytest.mark.parametrize("seed,C,H,W", [(0, 4, 40, 56), (1, 2, 17, 23), (2, 6, 64, 48), (3, 3, 33, 65), (4, 5, 29, 31),
                                        (5, 2, 100, 20), (6, 7, 24, 40)])
def test_unproject_and_clean_vs_oracle(seed, C, H, W):
    from starst3r_amd import ops
    ctx = ops.get_context(torch.device(DEV))
    P = make_problem(C, H, W, seed=seed)
    n = H * W
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(DEV)
    start = t(np.arange(C + 1) * n, torch.int32)
    pix = t(np.tile(P["pix"], (C, 1))); idx = t(np.concatenate(P["idxs"]), torch.int32); off = t(np.concatenate(P["offs"]))
    cam = t(P["cam"]); core = t(P["core"]); bf = t(P["base_f"])
    pts, z = ops.dense_unproject(ctx, start, pix, idx, off, core, cam, bf)
    pts_ref, z_ref = [], []
    for k in range(C):
        # float32 camera rows are what the kernel sees: restate with exactly those numbers
        K = np.array([[P["cam"][k, 12], 0, P["cam"][k, 13]], [0, P["cam"][k, 12], P["cam"][k, 14]], [0, 0, 1]], np.float64)
        c2w = np.eye(4); c2w[:3, :3] = P["cam"][k, :9].reshape(3, 3); c2w[:3, 3] = P["cam"][k, 9:12]
        dm = P["A"][k].astype(np.float64) + P["B"][k].astype(np.float64) * P["core"][k].astype(np.float64)
        a, b = do.unproject(P["pix"].astype(np.float64), P["idxs"][k], P["offs"][k], dm, K, c2w, P["base_f"][k])
        pts_ref.append(a); z_ref.append(b)
    np.testing.assert_allclose(pts.cpu().numpy(), np.concatenate(pts_ref), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(z.cpu().numpy(), np.concatenate(z_ref), rtol=1e-5)
    # clean: feed the oracle the kernel's own float32 points/depths so only the decisions are compared
    conf = t(np.concatenate(P["confs"]))
    sizes = t(np.array([[H, W]] * C), torch.int32)
    got = ops.dense_clean(ctx, start, sizes, cam, pts, z, conf).cpu().numpy().reshape(C, n)
    ptsn = pts.cpu().numpy().astype(np.float64).reshape(C, n, 3); zn = z.cpu().numpy().astype(np.float64).reshape(C, n)
    Ks = [np.array([[P["cam"][k, 12], 0, P["cam"][k, 13]], [0, P["cam"][k, 12], P["cam"][k, 14]], [0, 0, 1]], np.float64)
          for k in range(C)]
    c2ws = []
    for k in range(C):
        m = np.eye(4); m[:3, :3] = P["cam"][k, :9].reshape(3, 3); m[:3, 3] = P["cam"][k, 9:12]; c2ws.append(m)
    margin = []
    want = do.clean_pointcloud(P["confs"], Ks, c2ws, list(zn), list(ptsn), [(H, W)] * C, margin=margin)
    want = np.stack(want); margin = np.stack(margin)
    lowered = (want < np.stack(P["confs"]))
    assert lowered.mean() > 0.005                                  # the fliers are caught
    safe = margin > 2e-3                                           # away from rounding / depth-test knife edges
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(got[safe], want[safe].astype(np.float32))
    assert (got != want.astype(np.float32)).mean() < 0.01         # and hardly any disagreement on the rest either
    assert (got <= np.stack(P["confs"])).all()


def test_scene_seeding_uses_cleaned_confidences():
    import starst3r_amd as st
    from st3r_synth.synth_model import SyntheticPairwiseModel
    model = SyntheticPairwiseModel(width=128, height=96, n_corr=300, seed=2)
    sc = st.Scene(device=DEV)
    sc.add_images(model, [torch.zeros(3, 96, 128) for _ in range(3)])
    n = sum(p.shape[0] for p in sc.dense_pts)
    assert 0 < n < 3 * 96 * 128 and all(p.shape[1] == 3 for p in sc.dense_pts)
    assert sc.dense_pts_flat.shape == sc.dense_cols_flat.shape
    # the synthetic scene is a unit sphere: seeded points lie on it (in the optimiser's gauge up to a similarity,
    # so only the spread of the radius around its mean is checked)
    P = sc.dense_pts_flat.double().numpy()
    A = np.concatenate([2 * P, np.ones((len(P), 1))], 1)               # |p|^2 = 2 p.c + (r^2 - |c|^2)
    sol, *_ = np.linalg.lstsq(A, (P * P).sum(1), rcond=None)
    c = sol[:3]; r = np.sqrt(sol[3] + c @ c)
    res = np.linalg.norm(P - c, axis=1) - r
    assert np.abs(res).mean() / r < 0.05, (np.abs(res).mean(), r)
