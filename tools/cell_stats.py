"""Offline estimate (numpy, CPU) of what per-cell record lists would save in the blend loops at SYNTH-1M.

For a sample of 16x16 tiles of one SYNTH-1M view it evaluates alpha >= 1/255 for every (record, pixel) of the tile and
reports, per forward batch of 256 depth-sorted records:
  quadrant trips   = sum over the four 8x8 quadrants of the records that reach the quadrant (today's formulation)
  cell trips       = sum over the four waves of max over the wave's four 4x4 cells of the records reaching the cell
                     (four independent lists per wave, the wave runs until its longest list is done)
  ideal cell trips = sum over cells of the list lengths / 4 (no imbalance)
and the lane utilisation of each formulation.  python tools/cell_stats.py [n_tiles] [view]
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from st3r_synth import synth  # noqa: E402


def quat_to_rot(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def project(g, w2c, K, W, H):
    R, t = w2c[:3, :3].astype(np.float64), w2c[:3, 3].astype(np.float64)
    p = g["means"].astype(np.float64) @ R.T + t
    z = p[:, 2]
    ok = z > 0.01
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Rq = quat_to_rot(g["quats"].astype(np.float64))
    M = Rq * g["scales"].astype(np.float64)[:, None, :]
    Sig = M @ M.transpose(0, 2, 1)
    Sc = R @ Sig @ R.T
    limx, limy = 1.3 * 0.5 * W / fx, 1.3 * 0.5 * H / fy
    tx = z * np.clip(p[:, 0] / z, -limx, limx); ty = z * np.clip(p[:, 1] / z, -limy, limy)
    J = np.zeros((len(z), 2, 3))
    J[:, 0, 0] = fx / z; J[:, 0, 2] = -fx * tx / z ** 2
    J[:, 1, 1] = fy / z; J[:, 1, 2] = -fy * ty / z ** 2
    S2 = J @ Sc @ J.transpose(0, 2, 1)
    S2[:, 0, 0] += 0.3; S2[:, 1, 1] += 0.3
    det = S2[:, 0, 0] * S2[:, 1, 1] - S2[:, 0, 1] ** 2
    ca, cb, cc = S2[:, 1, 1] / det, -S2[:, 0, 1] / det, S2[:, 0, 0] / det
    mx, my = fx * p[:, 0] / z + cx, fy * p[:, 1] / z + cy
    return ok, mx, my, z, ca, cb, cc


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    W, H = 1920, 1080
    g, w2c, Ks = synth.make_scene(1_000_000, 8, W, H)
    ok, mx, my, z, ca, cb, cc = project(g, w2c[view], Ks[view], W, H)
    op = g["opacities"].astype(np.float64)
    tau = np.log(np.maximum(255.0 * op, 1e-30))
    ok &= tau > 0
    det = ca * cc - cb * cb
    ex = np.sqrt(np.maximum(2 * tau * cc / det, 0)); ey = np.sqrt(np.maximum(2 * tau * ca / det, 0))
    rng = np.random.default_rng(1)
    tw, th = W // 16, (H + 15) // 16
    tot = dict(records=0, quad=0, cell=0, cell_ideal=0, pix_hits=0, cell_pairs=0, half=0, batches=0, quad_lanes=0, c2=0, c2_ideal=0)
    order = np.argsort(z, kind="stable")
    for _ in range(n_tiles):
        tx_, ty_ = rng.integers(0, tw), rng.integers(0, th)
        x0, y0 = 16 * tx_, 16 * ty_
        sel = ok & (mx + ex >= x0 + 0.5) & (mx - ex <= x0 + 15.5) & (my + ey >= y0 + 0.5) & (my - ey <= y0 + 15.5)
        ids = order[sel[order]]
        if len(ids) == 0:
            continue
        px = x0 + 0.5 + np.arange(16); py = y0 + 0.5 + np.arange(16)
        dx = mx[ids][:, None, None] - px[None, None, :]
        dy = my[ids][:, None, None] - py[None, :, None]
        sig = 0.5 * (ca[ids][:, None, None] * dx * dx + cc[ids][:, None, None] * dy * dy) + cb[ids][:, None, None] * dx * dy
        hit = (sig <= tau[ids][:, None, None]) & (py[None, :, None] < H)   # [n, y, x]
        keep = hit.any(axis=(1, 2))
        hit = hit[keep]
        n = hit.shape[0]
        # transmittance: a pixel stops at the first record that would take T to <= 1e-4 (that record is not blended)
        alpha = np.where(hit, np.minimum(0.999, op[ids][keep][:, None, None] * np.exp(-sig[keep])), 0.0)
        T = np.ones((16, 16)); live = np.ones((16, 16), bool)
        live_at = np.zeros((n, 16, 16), bool)       # pixel still live when record k arrives
        for k in range(n):
            live_at[k] = live
            nT = T * (1 - alpha[k])
            stop = live & (nT <= 1e-4)
            live &= ~stop
            T = np.where(live, nT, T)
        tile_live = live_at.any(axis=(1, 2))
        # batches the tile consumes (the kernel checks once per 256 records)
        nb = 0
        while nb * 256 < n and tile_live[nb * 256]:
            nb += 1
        n = min(n, nb * 256)
        hit = hit[:n]; live_at = live_at[:n]
        useful = hit & live_at
        tot["useful"] = tot.get("useful", 0) + int(useful.sum())
        qlive = live_at.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).reshape(n, 4)
        # the wave checks its live pixels once per 64-record chunk
        qlive_chunk = qlive[(np.arange(n) // 64) * 64]
        tot["records"] += n
        tot["pix_hits"] += int(hit.sum())
        quad = hit.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).reshape(n, 4)             # [n, quadrant]
        cell = hit.reshape(n, 4, 4, 4, 4).any(axis=(2, 4))                            # [n, cy, cx]
        tot["quad"] += int((quad & qlive_chunk).sum())
        tot["quad_all"] = tot.get("quad_all", 0) + int(quad.sum())
        clive = live_at.reshape(n, 4, 4, 4, 4).any(axis=(2, 4))                       # cell has a live pixel at record k
        tot["cell_pairs"] += int(cell.sum())
        # 2x2-pixel cells (16 lists per wave)
        c2 = hit.reshape(n, 8, 2, 8, 2).any(axis=(2, 4))                              # [n, 8, 8]
        for b0 in range(0, n, 256):
            cb_ = cell[b0:b0 + 256] & clive[b0][None]      # cells already dead at the start of the batch get empty lists
            cb_nd = cell[b0:b0 + 256]
            tot["batches"] += 1
            for w in range(4):
                wy, wx = w >> 1, w & 1
                if not qlive[b0, w]:
                    continue
                lens = cb_[:, 2 * wy:2 * wy + 2, 2 * wx:2 * wx + 2].sum(axis=0).ravel()
                tot["cell"] += int(lens.max())
                # variant: the wave re-checks its live pixels every 32 trips (list positions) and stops when none is left
                sub = cb_[:, 2 * wy:2 * wy + 2, 2 * wx:2 * wx + 2].reshape(-1, 4)        # [records of batch, 4 cells]
                posn = np.cumsum(sub, axis=0)                                             # list position after each record
                nmax = int(lens.max()); done = 0
                for k0 in range(0, nmax, 32):
                    # record index at which every cell's list has reached position k0 (the slowest list decides nothing:
                    # a window starts when the wave's trip counter reaches k0, i.e. each row is at its own k0-th entry)
                    recs = [int(np.searchsorted(posn[:, c], k0, side="left")) for c in range(4) if lens[c] > k0]
                    r0 = min(recs) if recs else len(sub)
                    alive = qlive[min(b0 + r0, n - 1), w]
                    if not alive:
                        break
                    done = min(k0 + 32, nmax)
                tot["cell_win"] = tot.get("cell_win", 0) + done
                tot["cell_nd"] = tot.get("cell_nd", 0) + int(cb_nd[:, 2 * wy:2 * wy + 2, 2 * wx:2 * wx + 2].sum(axis=0).max())
                tot["cell_ideal"] += lens.sum() / 4.0
                l2 = c2[b0:b0 + 256, 4 * wy:4 * wy + 4, 4 * wx:4 * wx + 4].sum(axis=0).ravel()
                tot["c2"] += int(l2.max()); tot["c2_ideal"] += l2.sum() / 16.0
    r = tot["records"]
    print(f"tiles sampled {n_tiles}, records {r} ({r / n_tiles:.0f} per tile), pixel hits per record {tot['pix_hits'] / r:.1f}")
    print(f"quadrant trips {tot['quad']} ({tot['quad'] / r:.3f} per record)  lane utilisation {tot['pix_hits'] / (64.0 * tot['quad']):.3f}")
    print(f"(record, cell) pairs {tot['cell_pairs']} ({tot['cell_pairs'] / r:.3f} per record)")
    print(f"cell trips (max of 4 per 256-batch) {tot['cell']}  = {tot['cell'] / tot['quad']:.3f} of the quadrant trips; "
          f"ideal {tot['cell_ideal']:.0f} = {tot['cell_ideal'] / tot['quad']:.3f}; lane utilisation {tot['pix_hits'] / (64.0 * tot['cell']):.3f}")
    print(f"useful (record, pixel) pairs {tot['useful']}: utilisation quadrant {tot['useful'] / (64.0 * tot['quad']):.3f} cells {tot['useful'] / (64.0 * tot['cell']):.3f}; "
          f"cells without dead-cell pruning {tot['cell_nd'] / tot['quad']:.3f} of quadrant trips; quadrant trips ignoring saturation {tot['quad_all']}")
    print(f"cells with a liveness check every 32 trips: {tot['cell_win'] / tot['quad']:.3f} of the quadrant trips")
    print(f"2x2 cells: trips {tot['c2']} = {tot['c2'] / tot['quad']:.3f} of quadrant trips; ideal {tot['c2_ideal'] / tot['quad']:.3f}")
    print(f"batches {tot['batches']}, trips per wave per batch: quadrant {tot['quad'] / 4.0 / tot['batches']:.1f}, cells {tot['cell'] / 4.0 / tot['batches']:.1f}")


if __name__ == "__main__":
    main()
