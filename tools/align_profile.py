"""Where the alignment spends its time (library built with ST3R_DEFS=-DALIGN_PROFILE), shader-clock ticks:
  * sections of the update phase (k_align_update, default two-launch form), thread 0 between its barriers;
  * ST3R_DEBUG_FLAGS=1024: phases of the persistent kernel, workgroup 0 per iteration: residual | barrier | update | barrier.
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
if not (int(os.environ.get("ST3R_DEBUG_FLAGS", "0")) & 1024):
    sys.exit(0)
buf = (ctypes.c_ulonglong * 4)()
L.st3r_debug_align_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.st3r_debug_align_profile(ctx.handle, ctypes.cast(buf, ctypes.c_void_p))
tot = sum(buf)
print(f"views {views}, rows {len(flat['corr_a1'])} / {len(flat['c2d_a2'])} (+{len(flat['dust_a1'])}), wall {dt * 1e3:.2f} ms = {dt / 700 * 1e6:.1f} us per iteration")
for n, v in zip(("residual phase", "barrier behind it", "update phase", "barrier behind it"), buf):
    print(f"  {n:20s} {v / 700:10.0f} ticks per iteration  {100.0 * v / tot:5.1f} %  ~ {dt / 700 * 1e6 * v / tot:5.1f} us")
