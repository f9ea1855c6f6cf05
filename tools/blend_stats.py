"""Workload statistics of the blend forward (library must be built with ST3R_DEFS=-DST3R_STATS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starst3r_amd import ops, _lib
from st3r_synth import synth
ctx = ops.get_context("cuda:0")
N, V, W, H = 1_000_000, 8, 1920, 1080
g, w2c, Ks = synth.make_scene(N, V, W, H)
P = {k: torch.tensor(v, device="cuda:0") for k, v in g.items()}
vm, K = torch.tensor(w2c, device="cuda:0"), torch.tensor(Ks, device="cuda:0")
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
L.st3r_debug_blend_stats(None, 1)
rgb, alpha, st = ops.render(ctx, P, vm, K, ops.camera_positions(vm), W, H)
torch.cuda.synchronize()
L.st3r_debug_blend_stats(buf, 0)
I = st["n_isects"]
names = ["batches", "relevant (entry,wave)", "any-valid", "contributed", "valid lanes", "taken lanes", "entries in lists", "staged records relevant to >=1 quadrant"]
for n, v in zip(names, buf):
    print(f"{n:24s} {v:>14d}")
print("I =", I, " entries staged =", buf[0] * 256, " staged/I = %.3f" % (buf[0] * 256 / I))
print("relevant / (staged*4) = %.3f" % (buf[1] / (buf[0] * 256 * 4)))
print("any-valid / relevant = %.3f, contributed / relevant = %.3f" % (buf[2] / buf[1], buf[3] / buf[1]))
print("valid lanes per any-valid iter = %.1f ; taken lanes per contributed iter = %.1f" % (buf[4] / buf[2], buf[5] / buf[3]))
print("taken lanes per pixel = %.1f" % (buf[5] / (V * W * H)))
