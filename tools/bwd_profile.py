"""Work counts of k_blend_bwd (library built with ST3R_DEFS=-DBWD_PROFILE): rounds, record trips and phase-2 passes of one
fused train step on SYNTH-1M -- the numbers DESIGN.md's instruction budget of the kernel is built on.  The build also
timestamps the phases of a round (s_memtime), but the timestamps themselves slow the kernel down 6x (each waits for the
scalar-memory counter, which the LDS operations share), so the phase shares it prints are NOT those of the production
kernel; they are kept as a diagnostic of the instrumented build only.  python tools/bwd_profile.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starst3r_amd import ops, _lib
from st3r_synth import synth
dev = torch.device("cuda:0")
ctx = ops.get_context(dev)
N, V, W, H = 1_000_000, 8, 1920, 1080
g, w2c, Ks = synth.make_scene(N, V, W, H)
P = {k: torch.tensor(v, device=dev) for k, v in g.items()}
vm, K = torch.tensor(w2c, device=dev), torch.tensor(Ks, device=dev)
gt_g = synth.perturb_for_gt(g)
Q = {k: torch.tensor(v, device=dev) for k, v in gt_g.items()}
gt, _, _ = ops.render(ctx, Q, vm, K, ops.camera_positions(vm), W, H)
gt = gt.clamp(0, 1).contiguous()
grads = torch.empty(23 * N, device=dev); m = torch.zeros_like(grads); v = torch.zeros_like(grads); loss = torch.zeros(1, device=dev)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize(); L.st3r_debug_bwd_profile(None, 1); e0.record()
    ops.train_step(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 0.0, 0.9, 0.999, 1e-8, it + 1, loss)
e1.record(); torch.cuda.synchronize()
print(f"whole step of the instrumented build: {e0.elapsed_time(e1):.2f} ms (production: ~5.3; the timestamps cost time, read the shares)")
L.st3r_debug_bwd_profile(buf, 0)
b = list(buf)
rounds_w = b[8]; rounds = rounds_w / 4
names = ["staging + barrier", "walk (phases 1 + 2)", "wait behind the walk", "flush"]
print(f"rounds {rounds:.0f} (x4 waves), record trips {b[9]} ({b[9] / rounds_w:.1f} per wave and round), phase-2 passes {b[10]} ({b[10] / rounds_w:.2f})")
for o, tag, nw in ((0, "wave 0   ", 1), (4, "waves 1-3", 3)):
    tot = sum(b[o:o + 4])
    print(tag, "cycles per round:", "  ".join(f"{n} {b[o + k] / (rounds * nw):.0f}" for k, n in enumerate(names)), f"  = {tot / (rounds * nw):.0f}")
