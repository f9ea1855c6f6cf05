"""CPU restatement (numpy) of the MCMC refinement hooks -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(starst3r_amd/) never does.

What it restates: gsplat.MCMCStrategy.step_post_backward with default hyper-parameters, as reached from
the reference at starster/gs.py:43-45 (construction) and starster/gs.py:163-164 (the call, lr = 1e-3 literal).
gsplat is a third-party dependency that is absent from /root/reference (requirements.txt:1, unpinned; the
1.4.x line at the reference date); the algorithm below follows SURVEY.md App. A.2:

  relocate   dead = sigmoid(opacities) <= min_opacity; one source per dead Gaussian drawn with replacement
             with probability ~ sigmoid(opacity) among the alive; ratio = (times drawn) + 1 clamped to [1, 51];
             compute_relocation; clamp(new_opacity, min_opacity, 1 - eps) -> logit; new_scale -> log;
             dead rows copy all parameters of their source; Adam moments of the sources are zeroed
  add_new    n_new = min(cap_max, int(1.05 N)) - N draws among all Gaussians, same relocation maths, the
             copies are appended, optimiser state is zero-extended (sources keep theirs)
  noise      means += Sigma(quats, exp(scales)) @ (randn * sigmoid_100(1 - sigmoid(o) - 0.995) * lr * noise_lr)

PARITY UNPINNED: the reference has no test or golden vector for this path and torch.multinomial /
torch.randn streams cannot be reproduced.  The build draws from Philox4x32-10 (key = seed, counter =
(index, stream, step)) with exact integer inverse-CDF sampling; this file replays those draws bit for bit.
The Philox implementation itself is pinned against the published Random123 known-answer vectors
(tests/test_oracle_mcmc.py).
"""
import math

import numpy as np

NMAX = 51
STREAM_RELOCATE, STREAM_ADD, STREAM_NOISE = 0, 1, 2
F32_EPS = np.float32(1.1920929e-07)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds (Salmon et al., SC'11); all arguments broadcastable uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(k0) & np.uint64(0xFFFFFFFF); k1 = np.uint64(k1) & np.uint64(0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF); sh = np.uint64(32)
    for _ in range(10):
        p0 = M0 * c0; p1 = M1 * c2
        n0 = (p1 >> sh) ^ c1 ^ k0
        n1 = p1 & mask
        n2 = (p0 >> sh) ^ c3 ^ k1
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask; k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def binom_table():
    """float32 [51,51] table of C(n,k) (gsplat MCMCStrategy.initialize_state)."""
    t = np.zeros((NMAX, NMAX), np.float32)
    for n in range(NMAX):
        for k in range(n + 1):
            t[n, k] = math.comb(n, k)
    return t


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def weights(opacities, min_opacity, relocating):
    """24-bit fixed-point sampling weights and the dead mask."""
    p = sigmoid32(opacities)
    dead = (p <= np.float32(min_opacity)) if relocating else np.zeros(p.shape, bool)
    w = np.where(dead, 0, np.floor(p.astype(np.float64) * 16777216.0)).astype(np.uint64)
    return w, dead


def draw(cum, idx, stream, step, seed):
    """Source index of draws `idx` given the inclusive prefix sums `cum` (uint64)."""
    idx = np.asarray(idx, np.uint64)
    r0, r1, _, _ = philox4x32_10(idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32), stream, step,
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    total = int(cum[-1])
    r64 = [(int(a) << 32) | int(b) for a, b in zip(r0, r1)]
    t = np.array([(r * total) >> 64 for r in r64], dtype=np.uint64)
    out = np.searchsorted(cum, t, side="right")
    return np.minimum(out, len(cum) - 1).astype(np.int64)


def compute_relocation(o, s, ratio, binoms=None):
    """gsplat compute_relocation (float32): o [n] in (0,1), s [n,3] linear scales, ratio [n] int >= 1."""
    binoms = binom_table() if binoms is None else binoms
    o = np.asarray(o, np.float32); s = np.asarray(s, np.float32)
    new_o = np.empty_like(o); new_s = np.empty_like(s)
    for j in range(o.shape[0]):
        n_idx = int(min(max(int(ratio[j]), 1), NMAX))
        no = np.float32(1) - np.power(np.float32(1) - o[j], np.float32(1.0) / np.float32(n_idx), dtype=np.float32)
        denom = np.float32(0)
        for i in range(1, n_idx + 1):
            for k in range(i):
                term = np.float32((-1.0) ** k / math.sqrt(k + 1)) * np.power(no, np.float32(k + 1), dtype=np.float32)
                denom = np.float32(denom + binoms[i - 1, k] * term)
        new_o[j] = no
        new_s[j] = (o[j] / denom) * s[j]
    return new_o, new_s


def _apply_sources(P, src_unique, counts, min_opacity):
    o = sigmoid32(P["opacities"][src_unique])
    s = np.exp(P["scales"][src_unique], dtype=np.float32)
    new_o, new_s = compute_relocation(o, s, counts + 1)
    new_o = np.clip(new_o, np.float32(min_opacity), np.float32(1) - F32_EPS)
    P["opacities"][src_unique] = np.log(new_o / (np.float32(1) - new_o), dtype=np.float32)
    P["scales"][src_unique] = np.log(new_s, dtype=np.float32)


def relocate(P, adam=None, min_opacity=0.005, seed=0, step=0, cum_override=None):
    """In place on the dict of float32 arrays P (means, quats, scales, opacities, sh0, shN) and on
    adam = {key: (m, v)} arrays shaped like the parameters.  Returns (dead_ids, sampled)."""
    w, dead = weights(P["opacities"], min_opacity, True)
    cum = np.cumsum(w, dtype=np.uint64) if cum_override is None else cum_override
    dead_ids = np.nonzero(dead)[0]
    n = len(dead_ids)
    if n == 0 or cum[-1] == 0:
        return dead_ids, np.zeros(0, np.int64)
    sampled = draw(cum, np.arange(n), STREAM_RELOCATE, step, seed)
    uniq, counts = np.unique(sampled, return_counts=True)
    _apply_sources(P, uniq, counts, min_opacity)
    for k in P:
        P[k][dead_ids] = P[k][sampled]
    if adam is not None:
        for k, (m, v) in adam.items():
            m[uniq] = 0; v[uniq] = 0
    return dead_ids, sampled


def add_new(P, n_new, min_opacity=0.005, seed=0, step=0, cum_override=None):
    """Returns (new dict with n_new appended rows, sampled)."""
    w, _ = weights(P["opacities"], min_opacity, False)
    cum = np.cumsum(w, dtype=np.uint64) if cum_override is None else cum_override
    sampled = draw(cum, np.arange(n_new), STREAM_ADD, step, seed)
    uniq, counts = np.unique(sampled, return_counts=True)
    _apply_sources(P, uniq, counts, min_opacity)
    return {k: np.concatenate([v, v[sampled]]) for k, v in P.items()}, sampled


def normals(n, step, seed):
    """[n,3] standard normals of the noise stream (Box-Muller on 24-bit uniforms), evaluated in float64."""
    r = philox4x32_10(np.arange(n, dtype=np.uint64), 0, STREAM_NOISE, step, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for x in r]
    u = [x.astype(np.float64) for x in u]
    ra = np.sqrt(-2.0 * np.log(u[0])); rb = np.sqrt(-2.0 * np.log(u[2]))
    return np.stack([ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]), rb * np.cos(2 * np.pi * u[3])], -1)


def noise_delta(quats, scales, opacities, scaler, step, seed):
    """The increment inject_noise_to_position adds to the means (float64)."""
    q = np.asarray(quats, np.float64); q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    M = R * np.exp(np.asarray(scales, np.float64))[:, None, :]
    cov = M @ M.transpose(0, 2, 1)
    op = 1.0 / (1.0 + np.exp(-np.asarray(opacities, np.float64)))
    gate = 1.0 / (1.0 + np.exp(-100.0 * ((1.0 - op) - 0.995)))
    nz = normals(len(op), step, seed) * gate[:, None] * scaler
    return (cov @ nz[..., None])[..., 0]
