"""The record-level test behind the blend kernels' lean rounds / batches (gs_blend.hip, gs_blend_cells.hip, round 5):
a record whose conic has a > 0, c > 0 and a c - b^2 >= 2e-3 a c never yields P > 0 (gsplat's sigma < 0) in the kernels'
float evaluation, so the per-pixel test may be dropped for it.  Checked here by brute force on the CPU with the kernels' own
operation sequence (blend_common.h: blend_power; fma emulated through float64, whose 53 bits hold a float product
exactly), at the margin and along the direction where the quadratic form is smallest."""
import numpy as np

LOG2E = np.float32(1.4426950408889634)


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def blend_power(dx, dy, qa, qb, qc):
    """lx = fma(dx, qa, qb * dy); P = fma(dx, lx, (qc * dy) * dy) -- every step rounded to float32."""
    lx = fma32(dx, qa, f32(qb * dy))
    return fma32(dx, lx, f32(f32(qc * dy) * dy))


def q_form(a, b, c):
    """The staged record: (-0.5 log2e a, -log2e b, -0.5 log2e c), rounded as the kernels round them."""
    return f32(f32(-0.5) * LOG2E * a), f32(-LOG2E * b), f32(f32(-0.5) * LOG2E * c)


def easy(a, b, c):
    """The staging threads' test (float32, same expression)."""
    det = f32(f32(a * c) - f32(b * b))
    return (a > 0) & (c > 0) & (det >= f32(np.float32(2e-3) * f32(a * c)))


def test_easy_records_never_have_positive_power():
    rng = np.random.default_rng(5)
    n = 4_000_000
    # conics over ten decades, correlation pushed against the margin: 1 - rho^2 in [2e-3, 1.5e-2] for most samples
    a = f32(10.0 ** rng.uniform(-7, 1, n)); c = f32(10.0 ** rng.uniform(-7, 1, n))
    one_minus = np.where(rng.random(n) < 0.8, rng.uniform(2e-3, 1.5e-2, n), rng.uniform(2e-3, 1.0, n))
    rho = np.sqrt(1.0 - one_minus) * rng.choice([-1.0, 1.0], n)
    b = f32(rho * np.sqrt(a.astype(np.float64) * c.astype(np.float64)))
    ok = easy(a, b, c)
    assert ok.mean() > 0.9            # (the float32 test rejects a few samples that sit exactly on the margin)
    a, b, c = a[ok], b[ok], c[ok]
    n = a.size
    qa, qb, qc = q_form(a, b, c)
    # offsets: pixel centres minus a mean -- magnitudes from 1e-3 to 4000 pixels; half of them along the conic's weakest
    # direction (where Q / (|qa| dx^2 + |qc| dy^2) is smallest: dx : dy = -sign(b) sqrt(c) : sqrt(a))
    r = f32(10.0 ** rng.uniform(-3, 3.6, n))
    th = rng.uniform(0, 2 * np.pi, n)
    dx = f32(r * np.cos(th)); dy = f32(r * np.sin(th))
    weak = rng.random(n) < 0.5
    wx = -np.sign(b) * np.sqrt(c.astype(np.float64)); wy = np.sqrt(a.astype(np.float64))
    wn = np.hypot(wx, wy)
    dx = np.where(weak, f32(r * wx / wn), dx); dy = np.where(weak, f32(r * wy / wn), dy)
    P = blend_power(f32(dx), f32(dy), qa, qb, qc)
    assert not np.any(P > 0), (int(np.sum(P > 0)), float(P.max()))
    # and the margin is not idle: just inside it the relative size of P against the positive-term sum is ~1e-3 -- three
    # thousand times the five roundings' 4 * 2^-24
    S = np.abs(qa.astype(np.float64)) * dx.astype(np.float64) ** 2 + np.abs(qc.astype(np.float64)) * dy.astype(np.float64) ** 2
    rel = -P.astype(np.float64)[S > 0] / S[S > 0]
    assert rel.min() > 2.5e-4, rel.min()


def test_records_outside_the_margin_are_flagged():
    """Degenerate, indefinite and NaN conics, and opacities above the clamp, take the full trip."""
    a = f32([1.0, 1.0, -1.0, 1.0, np.nan, 1.0, 0.0]); c = f32([1.0, 1.0, 1.0, -1.0, 1.0, 1.0, 1.0])
    b = f32([0.9995, 1.1, 0.0, 0.0, 0.0, np.nan, 0.0])
    assert not easy(a, b, c).any()
    assert easy(f32([1.0]), f32([0.9]), f32([1.0])).all()


def test_alpha_threshold_as_arithmetic_is_exact():
    """clamp01((alpha - t') 2^64) with t' the float below 1/255 is the step function of `alpha >= 1/255` (the easy backward
    rounds: one multiply-add with the clamp modifier and one multiply instead of compare + select)."""
    t = np.float32(1.0) / np.float32(255.0)
    assert t.view(np.uint32) == 0x3B808081
    tp = np.nextafter(t, np.float32(0))
    assert tp.view(np.uint32) == 0x3B808080      # the constant of gs_blend.hip
    big = np.float32(2.0 ** 64)
    rng = np.random.default_rng(1)
    around = (t.view(np.uint32).astype(np.int64) + np.arange(-2000, 2001)).astype(np.uint32).view(np.float32)
    al = np.concatenate([around, f32(rng.uniform(0, 1.2, 200000)), f32(10.0 ** rng.uniform(-30, 0, 200000)), f32([0.0, 1.0, 0.999])])
    step = np.clip(fma32(al, np.full_like(al, big), np.full_like(al, -tp * big)), 0.0, 1.0)
    expect = (~(al < t)).astype(np.float32)
    assert np.array_equal(step, expect)
    assert np.array_equal(f32(al * step), np.where(al < t, np.float32(0), al))
