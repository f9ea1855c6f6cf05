"""Fraction of saturated pixels (T <= 1e-4) in the SYNTH-1M bench scene: decides whether per-wave early exits in the
blend forward could matter.  python tools/saturation.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from starst3r_amd import ops
from st3r_synth import synth

N, V, W, H = 1_000_000, 8, 1920, 1080
g, w2c, Ks = synth.make_scene(N, V, W, H, seed=0)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
ctx = ops.get_context(torch.device("cuda:0"))
P = {k: dev(v) for k, v in g.items()}
vm, K = dev(w2c), dev(Ks)
rgb, alpha, st = ops.render(ctx, P, vm, K, ops.camera_positions(vm), W, H)
a = alpha.reshape(V, H, W)
print("stats", st)
print("pixels with alpha > 0.9999:", float((a > 0.9999).float().mean()), " mean alpha", float(a.mean()))
t = (a > 0.9999).reshape(V, H // 8, 8, W // 8, 8).float().mean(dim=(2, 4))
print("8x8 quadrants fully saturated:", float((t == 1.0).float().mean()))
