"""CPU checks of the MCMC-refinement oracle (oracle/mcmc_oracle.py): the generator against the published
Random123 known-answer vectors, the relocation maths against closed forms, the sampler against its
target distribution."""
import math

import numpy as np

from oracle import mcmc_oracle as mo


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds: counter[4] key[2] -> output[4]
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = mo.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_binomial_table():
    t = mo.binom_table()
    assert t.shape == (51, 51) and t[0, 0] == 1 and t[5, 2] == 10 and t[4, 5] == 0
    assert t[50, 25] == np.float32(math.comb(50, 25))


def test_relocation_closed_forms():
    o = np.array([0.3, 0.9, 0.05], np.float32); s = np.array([[1, 2, 3]] * 3, np.float32)
    # ratio 1: nothing changes
    no, ns = mo.compute_relocation(o, s, np.array([1, 1, 1]))
    np.testing.assert_allclose(no, o, rtol=1e-6); np.testing.assert_allclose(ns, s, rtol=1e-6)
    # ratio 2: o' = 1 - sqrt(1 - o); denominator = 2 o' - o'^2 / sqrt(2)
    no, ns = mo.compute_relocation(o, s, np.array([2, 2, 2]))
    want_o = 1 - np.sqrt(1 - o.astype(np.float64))
    np.testing.assert_allclose(no, want_o, rtol=1e-5)
    want_s = s * (o / (2 * want_o - want_o ** 2 / math.sqrt(2)))[:, None]
    np.testing.assert_allclose(ns, want_s, rtol=1e-5)
    # the n copies composite back to the source opacity: 1 - (1 - o')^n = o
    for r in (3, 7):
        no, _ = mo.compute_relocation(o, s, np.array([r] * 3))
        np.testing.assert_allclose(1 - (1 - no.astype(np.float64)) ** r, o, rtol=1e-4)


def test_draws_follow_weights_and_skip_dead():
    rng = np.random.default_rng(1)
    op = np.clip(rng.normal(0, 2, 64), -4, 4).astype(np.float32)
    op[[3, 17, 40]] = -9.0  # dead: sigmoid < 0.005
    w, dead = mo.weights(op, 0.005, True)
    assert dead.sum() == 3 and (w[dead] == 0).all()
    cum = np.cumsum(w, dtype=np.uint64)
    n = 40000
    idx = mo.draw(cum, np.arange(n), mo.STREAM_RELOCATE, 7, 1234)
    assert not np.isin(idx, [3, 17, 40]).any()
    freq = np.bincount(idx, minlength=64)
    p = w / w.sum()
    alive = ~dead
    chi2 = (((freq - n * p) ** 2)[alive] / (n * p)[alive]).sum()
    assert chi2 < 120  # 60 dof: mean 60, sd 11
    # different step / stream / seed -> different draws; same arguments -> same draws
    assert (mo.draw(cum, np.arange(64), 0, 7, 1234) == idx[:64]).all()
    assert (mo.draw(cum, np.arange(64), 0, 8, 1234) != idx[:64]).any()
    assert (mo.draw(cum, np.arange(64), 1, 7, 1234) != idx[:64]).any()


def test_relocate_and_add_bookkeeping():
    rng = np.random.default_rng(2)
    N = 200
    P = {"means": rng.normal(size=(N, 3)), "quats": rng.normal(size=(N, 4)), "scales": rng.normal(-4, 0.3, (N, 3)),
         "opacities": np.clip(rng.normal(0, 2, N), -4, 4), "sh0": rng.normal(size=(N, 1, 3)), "shN": rng.normal(size=(N, 24, 3))}
    P = {k: v.astype(np.float32) for k, v in P.items()}
    P["opacities"][:20] = -8.0
    before = {k: v.copy() for k, v in P.items()}
    adam = {k: (np.ones_like(v), np.ones_like(v)) for k, v in P.items()}
    dead_ids, sampled = mo.relocate(P, adam, 0.005, seed=5, step=3)
    assert (dead_ids == np.arange(20)).all() and (sampled >= 20).all()
    untouched = np.setdiff1d(np.arange(N), np.concatenate([dead_ids, sampled]))
    for k in P:
        np.testing.assert_array_equal(P[k][untouched], before[k][untouched])
        np.testing.assert_array_equal(P[k][dead_ids], P[k][sampled])
    assert (mo.sigmoid32(P["opacities"][sampled]) < mo.sigmoid32(before["opacities"][sampled]) + 1e-6).all()
    for k, (m, v) in adam.items():
        assert (m[sampled] == 0).all() and (m[dead_ids] == 1).all() and (m[untouched] == 1).all()
    grown, s2 = mo.add_new(P, 10, 0.005, seed=5, step=3)
    assert grown["means"].shape[0] == N + 10
    np.testing.assert_array_equal(grown["shN"][N:], grown["shN"][s2])


def test_noise_stream_is_standard_normal():
    z = mo.normals(200000, 11, 99)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.corrcoef(z.T)[0, 1]) < 0.01 and abs(np.corrcoef(z.T)[0, 2]) < 0.01
