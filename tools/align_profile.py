"""Where the alignment spends its time (library built with ST3R_DEFS=-DALIGN_PROFILE), shader-clock ticks: sections of
the update phase (k_align_update), thread 0 between its barriers.  (Round 3 also profiled a persistent one-launch form of
the loop here; it measured no faster and was removed in round 4 -- tools/experiments/README.md keeps the numbers.)
python tools/align_profile.py [views]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starst3r_amd import ops, _lib, align
from st3r_synth import synth_align
views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
flat = synth_align.flatten(synth_align.make_problem(n_views=views, n_corr=2000 // (views - 1) + 1, seed=1))   # = bench.py's
ctx = ops.get_context("cuda:0")
L = ctypes.CDLL(_lib.LIB_PATH)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, par = align.run(flat, niter1=500, niter2=200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
ub = (ctypes.c_ulonglong * 8)()
L.st3r_debug_update_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.st3r_debug_update_profile(ctypes.cast(ub, ctypes.c_void_p), 0)
calls = 2 * 701
names = ["partial sums -> acc", "loss, moment reset", "forward pieces + chain (cached)", "camera gradients, reverse chain", "quaternion VJP + Adam", "forward: new camera table"]
ut = sum(ub[:6])
print(f"update phase, ticks per call (thread 0): total {ut / calls:.0f}")
for n, v in zip(names, ub):
    print(f"  {n:34s} {v / calls:9.0f}  {100.0 * v / max(ut, 1):5.1f} %")
print(f"views {views}: wall {dt * 1e3:.2f} ms = {dt / 700 * 1e6:.1f} us per iteration")
