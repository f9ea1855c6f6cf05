"""The product's `init_3dgs` / `run_3dgs_optim` / `render_3dgs_original` (starst3r_amd/gs.py over libst3r_hip.so) against
vectors produced by EXECUTING the reference's own /root/reference/starster/gs.py (tools/gen_gs_goldens.py ->
tests/golden/gs_loop_*.npz; `init_3dgs` :14-45, `compute_loss` :126-136, the loop :143-164).  The rasteriser / SSIM
underneath the reference run were this repository's oracle (gsplat / torchmetrics are absent: their arithmetic stays [U]);
what is pinned by reference EXECUTION is the reference-owned glue.  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("means", "scales", "quats", "opacities", "sh0", "shN")
DEV = "cuda:0"


def _scene(z):
    import starst3r_amd as st
    scene = st.Scene(device=DEV)
    scene.dense_pts = [torch.tensor(z["pts"])]
    scene.dense_cols = [torch.tensor(z["cols"])]
    scene.imgs = [z["imgs"][i] for i in range(z["imgs"].shape[0])]     # numpy (H, W, 3) in [0, 1], as Mast3r leaves them
    scene.c2w = torch.tensor(z["c2w"])
    scene.intrinsics = torch.tensor(z["Ks"])
    return scene


def _load(name):
    return np.load(os.path.join(GOLD, f"gs_loop_{name}.npz"))


@pytest.mark.parametrize("name", ["default", "args"])
def test_init_3dgs_equals_the_reference_run(name):
    z = _load(name)
    scene = _scene(z)
    scene.init_3dgs(float(z["init_scale"]), float(z["lr"]))
    assert tuple(scene.gaussians) == KEYS                                  # same keys, same order (gs.py:20-27)
    for k in KEYS:
        p = scene.gaussians[k]
        assert isinstance(p, torch.nn.Parameter) and p.device.type == "cuda" and p.dtype == torch.float32
        np.testing.assert_array_equal(p.detach().cpu().numpy(), z[f"init_{k}"], err_msg=k)
        assert scene.optimizers[k].param_groups[0]["lr"] == float(z["lr"])
    assert set(scene.optimizers) == set(KEYS) and scene.strategy is not None and scene.strategy_state is not None
    # the render the first iteration sees
    with torch.no_grad():
        img, alpha, info = scene.render_3dgs_original(int(z["W"]), int(z["H"]))
    img, alpha = img.cpu().numpy(), alpha.cpu().numpy()
    assert img.shape == z["render0"].shape and alpha.shape == z["alpha0"].shape
    # 1e-4 relative with the 1e-5 floor of DESIGN section 2; pixels whose include / stop decision float32 cannot decide
    # are bounded by 5e-3 (v_exp_f32 vs expf)
    err = np.abs(img - z["render0"]) / np.maximum(np.abs(z["render0"]), 0.1)
    assert np.mean(err <= 1e-4) >= 0.995 and err.max() <= 5e-3, (float(np.mean(err <= 1e-4)), float(err.max()))
    np.testing.assert_allclose(alpha, z["alpha0"], atol=5e-3)


@pytest.mark.parametrize("name", ["default", "args"])
def test_run_3dgs_optim_follows_the_reference_run(name):
    z = _load(name)
    scene = _scene(z)
    lr = float(z["lr"])
    scene.init_3dgs(float(z["init_scale"]), lr)
    fac = dict(loss_ssim_fac=float(z["loss_ssim_fac"]), loss_opacity_fac=float(z["loss_opacity_fac"]),
               loss_scale_fac=float(z["loss_scale_fac"]))
    losses, total, rows = [], 0, []
    for n in z["segments"]:
        ret = scene.run_3dgs_optim(int(n), **fac)
        assert isinstance(ret, list) and len(ret) == int(n) and all(isinstance(x, float) for x in ret)
        losses += ret; total += int(n)
        for k in KEYS:
            cur = scene.gaussians[k].detach().cpu().numpy(); ref = z[f"it{total}_{k}"]
            d = np.abs(cur - ref)
            rows.append((k, total, float(np.median(d)), float(np.percentile(d, 99.5)), float(d.max())))
            # "at the Adam-step scale": a step moves every element by ~lr whatever the size of its gradient, so elements
            # whose gradient is rounding noise may part by a fraction of lr per step; the bulk follows to float32 rounding
            # (measured on the saturating `args` scene: median 1.7e-6, p99.5 0.9 %, max 2.9 % of the lr * steps travelled;
            # round 6, whose loss kernel takes the ground-truth taps in one fixed association: p99.5 after the first TWO steps
            # 2.1 % -- the fifth largest of 1040 elements, each of them a noise-level gradient whose Adam step has the size lr)
            assert np.median(d) <= 5e-6, rows[-1]
            assert np.percentile(d, 99.5) <= 3e-2 * lr * total, rows[-1]
            assert d.max() <= 0.25 * lr * total, rows[-1]
        np.testing.assert_array_equal(scene.gaussians["sh0"].detach().cpu().numpy(), z["init_sh0"])
        np.testing.assert_array_equal(scene.gaussians["shN"].detach().cpu().numpy()[:, 4:], z["init_shN"][:, 4:])
        # optimiser state as the reference's six Adams hold it: none for sh0, a persisting step counter for the others
        assert scene.optimizers["sh0"].state == {} and int(z[f"it{total}_adam_has_state_sh0"]) == 0
        for k in ("means", "scales", "quats", "opacities", "shN"):
            st = next(iter(scene.optimizers[k].state.values()))
            assert st["step"] == total == float(z[f"it{total}_adam_step_{k}"])
        m = next(iter(scene.optimizers["means"].state.values()))["exp_avg"].cpu().numpy().reshape(-1, 3)
        ref_m = z[f"it{total}_adam_m_means"]
        assert np.abs(m - ref_m).max() <= 2e-3 * np.abs(ref_m).max()
    print("key, iterations, |param - reference run|: median, p99.5, max:", rows)
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-4)           # north_star: 1e-4 relative
    assert losses[-1] < 0.9 * losses[0]


class _Recorder:
    def __init__(self):
        self.log = []

    def check_sanity(self, params, optimizers):
        pass

    def initialize_state(self):
        return {}

    def step_pre_backward(self, params, optimizers, state, step, info):
        self.log.append((0, int(step), float("nan")))

    def step_post_backward(self, params, optimizers, state, step, info, lr):
        self.log.append((1, int(step), float(lr)))


def test_strategy_hooks_are_called_like_the_reference_calls_them():
    z = _load("hooks")
    scene = _scene(z)
    scene.init_3dgs()
    assert hasattr(scene.strategy, "step_pre_backward") and hasattr(scene.strategy, "step_post_backward")
    rec = _Recorder()
    scene.strategy = rec                                                   # a recorder, as in the reference run
    losses = []
    for n in z["segments"]:
        losses += scene.run_3dgs_optim(int(n), enable_pruning=True)
    assert [e[0] for e in rec.log] == z["hook_kind"].tolist()
    assert [e[1] for e in rec.log] == z["hook_step"].tolist()               # step restarts at 0 in every call
    np.testing.assert_array_equal([e[2] for e in rec.log][1::2], z["hook_lr"][1::2])   # the literal 1e-3
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-4)
