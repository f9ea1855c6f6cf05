"""The oracle's restatement of path C's reference-owned glue (oracle/gs_loop_oracle.py) against vectors produced by
EXECUTING /root/reference/starster/gs.py (tools/gen_gs_goldens.py -> tests/golden/gs_loop_*.npz): `init_3dgs` :14-45,
`compute_loss` :126-136, the loop :143-164.  CPU only; the reference itself is not needed at test time."""
import os

import numpy as np
import pytest

from oracle import gs_loop_oracle as glo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = glo.KEYS


def load(name):
    return np.load(os.path.join(GOLD, f"gs_loop_{name}.npz"))


@pytest.mark.parametrize("name", ["default", "args"])
def test_init_3dgs_parameter_dictionary(name):
    z = load(name)
    g = glo.init_params(z["pts"], z["cols"], float(z["init_scale"]))
    for k in KEYS:
        assert g[k].dtype == np.float32 and g[k].shape == z[f"init_{k}"].shape, k
        np.testing.assert_array_equal(g[k], z[f"init_{k}"], err_msg=k)
    # what the reference's own run says about the dictionary (gs.py:20-37)
    assert z["init_is_parameter"].all()
    np.testing.assert_allclose(z["init_lr_of_optimizers"], float(z["lr"]))
    np.testing.assert_array_equal(z["init_quats"][:, 0], 1.0)                     # w first
    np.testing.assert_array_equal(z["init_opacities"], 1.0)                       # raw, not a logit
    np.testing.assert_array_equal(z["init_scales"], np.float32(z["init_scale"]))  # raw, not a log
    for r in range(24):
        np.testing.assert_array_equal(z["init_shN"][:, r], 1 - z["cols"])
    for k in KEYS:
        assert int(z[f"init_adam_has_state_{k}"]) == 0


@pytest.mark.parametrize("name", ["default", "args"])
def test_loop_reproduces_the_reference_run(name):
    z = load(name)
    imgs = [z["imgs"][i] for i in range(z["imgs"].shape[0])]
    w2c = np.linalg.inv(z["c2w"].astype(np.float64)).astype(np.float32)
    W, H, lr = int(z["W"]), int(z["H"]), float(z["lr"])
    loop = glo.TrainLoop(glo.init_params(z["pts"], z["cols"], float(z["init_scale"])), lr)
    losses, total = [], 0
    for n in z["segments"]:
        for _ in range(int(n)):
            losses.append(loop.step(imgs, w2c, z["Ks"], W, H, float(z["loss_ssim_fac"]), float(z["loss_opacity_fac"]),
                                    float(z["loss_scale_fac"])))
        total += int(n)
        cur = loop.numpy()
        for k in KEYS:
            ref = z[f"it{total}_{k}"]
            d = np.abs(cur[k] - ref)
            # Adam turns a gradient into a step of ~lr whatever its size: an element whose gradient is rounding noise
            # (|g| ~ eps = 1e-8) may move differently; everything else follows to float32 rounding
            # (measured: median <= 1.2e-7, 99.5 % within 0.6 % and all within 1.3 % of the distance lr * steps travelled)
            assert np.median(d) <= 2e-7 and np.percentile(d, 99.5) <= 1e-2 * lr * total, (k, total, float(np.median(d)))
            assert d.max() <= 5e-2 * lr * total, (k, total, float(d.max()))
        # sh0 never moves and has no optimiser state; shN rows 4..23 receive zero gradients: state exists, values stay
        np.testing.assert_array_equal(cur["sh0"], z["init_sh0"])
        np.testing.assert_array_equal(cur["shN"][:, 4:], z["init_shN"][:, 4:])
        assert int(z[f"it{total}_adam_has_state_sh0"]) == 0 and int(z[f"it{total}_adam_has_state_shN"]) == 1
        assert float(z[f"it{total}_adam_step_shN"]) == total      # Adam's counters persist across calls (App. B-6)
        assert not loop.opt["sh0"].state
        m = next(iter(loop.opt["shN"].state.values()))["exp_avg"].numpy()
        np.testing.assert_allclose(m, z[f"it{total}_adam_m_shN"], atol=2e-3 * np.abs(m).max(), rtol=0)
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-5)   # measured 2.4e-7 / 5.0e-6
    assert losses[-1] < 0.9 * losses[0]


def test_hooks_and_rasterization_arguments_of_the_reference_run():
    z, d = load("hooks"), load("default")
    # the recorder changes nothing: same trajectory as without the hooks
    np.testing.assert_array_equal(z["losses"], d["losses"])
    seg = [int(n) for n in z["segments"]]
    steps = [s for n in seg for s in range(n)]                    # `step` restarts at 0 in every call (gs.py:143)
    kind, step, lr, aux = z["hook_kind"], z["hook_step"], z["hook_lr"], z["hook_aux"]
    assert kind.tolist() == [0, 1] * len(steps)                   # pre, post, pre, post, ...
    assert step[0::2].tolist() == steps and step[1::2].tolist() == steps
    np.testing.assert_array_equal(lr[1::2], 1e-3)                 # the literal of gs.py:164, whatever init_3dgs's lr
    assert aux[0::2].tolist() == list(range(1, len(steps) + 1))   # pre-backward runs after the iteration's render
    assert aux[1::2].all()                                        # post-backward runs after step() + zero_grad(None)
    assert z["setup_calls"].tolist() == [0, 1, 2]                 # MCMCStrategy(), check_sanity, initialize_state
    assert (z["raster_sh_degree"] == 1).all() and (z["raster_colors_rows"] == 24).all()   # colors=shN, sh_degree=1
    assert len(z["raster_sh_degree"]) == len(steps)               # ONE rasterization of all views per iteration
