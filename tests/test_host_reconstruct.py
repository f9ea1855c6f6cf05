"""Host-side restatement of the reference optimiser's set-up block (starster/reconstruct.py:148-207, 263-309):
the objects Mast3r's condense_data hands to sparse_scene_optimizer_slam -> the flat arrays of the C ABI.
The reference-format objects are the ones tools/gen_align_goldens.py feeds to the reference's own function."""
import importlib

import numpy as np
import pytest

from st3r_synth import synth_align as sa

rc = importlib.import_module("starst3r_amd.reconstruct")


@pytest.mark.parametrize("views,bad", [(2, False), (4, False), (4, True), (3, True)])
def test_reference_structures_flatten_to_the_same_arrays(views, bad):
    P = sa.make_problem(n_views=views, n_corr=150, seed=views, bad_pair=bad)
    a = sa.to_reference_inputs(P)
    got = rc.flatten_reference_inputs(a["imgs"], a["imsizes"], a["pps"], a["base_focals"], a["core_depth"], a["anchors"],
                                      a["corres"], a["corres2d"], a["preds_21"], a["mst"], matching_conf_thr=5)
    want = sa.flatten(P)
    assert set(got) == set(want)
    for k in want:
        x, y = np.asarray(got[k]), np.asarray(want[k])
        assert x.shape == y.shape and np.array_equal(x, y), k
    if bad:
        assert got["dust_a1"].size > 0      # the pair that fails `confs.max() > 5` went to the regression fallback
    else:
        assert got["dust_a1"].size == 0


def test_matching_gate_threshold_moves_pairs_between_losses():
    P = sa.make_problem(n_views=3, n_corr=100, seed=1)
    a = sa.to_reference_inputs(P)
    # no preds_21 for good pairs: raising the gate above every confidence must fail loudly on the missing fallback data
    with pytest.raises(KeyError):
        rc.flatten_reference_inputs(a["imgs"], a["imsizes"], a["pps"], a["base_focals"], a["core_depth"], a["anchors"],
                                    a["corres"], a["corres2d"], a["preds_21"], a["mst"], matching_conf_thr=1e9)


def test_unsupported_configurations_are_refused():
    P = sa.make_problem(n_views=2, n_corr=50, seed=0)
    a = sa.to_reference_inputs(P)
    args = (a["imgs"], 8, a["imsizes"], a["pps"], a["base_focals"], a["core_depth"], a["anchors"], a["corres"],
            a["corres2d"], a["preds_21"], None, a["mst"])
    with pytest.raises(NotImplementedError):
        rc.sparse_scene_optimizer_slam(*args, opt_depth=False, shared_intrinsics=True)
    with pytest.raises(NotImplementedError):
        rc.sparse_scene_optimizer_slam(*args, exp_depth=True)
    with pytest.raises(NotImplementedError):
        rc.sparse_scene_optimizer_slam(*args, depth_mode="mul")
    with pytest.raises(NotImplementedError):   # a loss the kernels cannot evaluate
        rc.sparse_scene_optimizer_slam(*args, loss1=lambda x, y: (x - y).abs().sum(-1))
    # the loss / schedule objects the kernels can take
    assert rc._gamma_of(rc.gamma_loss(0.5), 1.1) == 0.5 and rc._gamma_of(None, 1.1) == 1.1
    assert rc._gamma_of(rc.gamma_loss(1), 1.1) == 1.0 and rc._gamma_of(rc.l1_loss, 0.4) == 1.0
    assert rc.cosine_schedule(0.0, 0.07) == pytest.approx(0.07) and rc.cosine_schedule(1.0, 0.07) == pytest.approx(0.0)
    assert rc.linear_schedule(0.25, 0.08) == pytest.approx(0.06)


def test_gamma_of_mast3r_style_closures():
    """Mast3r's own gamma_loss closures -- free variables gamma / mul / offset / clip [U] -- incl. gamma = 1
    with an explicit zero offset (used to divide by zero) and callables whose repr merely contains "meta"."""
    def mast3r_gamma_loss(gamma, mul=1, offset=None, clip=float("inf")):
        if offset is None:
            offset = 0.0 if gamma == 1 else (1 / gamma) ** (1 / (gamma - 1))

        def loss_func(x, y):
            return (mul * rc.l1_loss(x, y).clip(max=clip) + offset) ** gamma - offset ** gamma
        return loss_func
    assert rc._gamma_of(mast3r_gamma_loss(1.1), 0.4) == pytest.approx(1.1)
    assert rc._gamma_of(mast3r_gamma_loss(1, offset=0.0), 0.4) == 1.0
    with pytest.raises(NotImplementedError):
        rc._gamma_of(mast3r_gamma_loss(1, offset=0.5), 0.4)
    with pytest.raises(NotImplementedError):
        rc._gamma_of(mast3r_gamma_loss(0.5, mul=2), 0.4)

    class metadata_loss:            # "meta" in the name, but an ordinary gamma loss object
        gamma = 0.7

        def __call__(self, x, y):
            return x
    assert rc._gamma_of(metadata_loss(), 1.1) == 0.7

    def meta_gamma_loss():          # the factory form the kernels cannot take
        return lambda alpha: rc.gamma_loss(1.1)
    f = meta_gamma_loss
    f.gamma = 1.1
    with pytest.raises(NotImplementedError):
        rc._gamma_of(f, 1.1)
