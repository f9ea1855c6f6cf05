import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from starst3r_amd import ops
from st3r_synth import synth
DEV="cuda:0"
ctx = ops.get_context(DEV)
n, v, w, h = 5_000_000, 8, 3840, 2160
g, w2c_np, Ks_np = synth.make_scene(n, v, w, h, seed=21)
P = {k: torch.tensor(val, device=DEV) for k, val in g.items()}
w2c = torch.tensor(w2c_np, device=DEV); Ks = torch.tensor(Ks_np, device=DEV)
campos = ops.camera_positions(w2c)
Q = {k: torch.tensor(val, device=DEV) for k, val in synth.perturb_for_gt(g).items()}
gt, _, st0 = ops.render(ctx, Q, w2c, Ks, campos, w, h)
gt = gt.clamp(0, 1).contiguous(); del Q
print("gt render isects", st0, flush=True)
grads = torch.empty(23 * n, device=DEV); m = torch.zeros_like(grads); vv = torch.zeros_like(grads)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 700
losses = torch.zeros(STEPS, device=DEV)
t0=time.time()
for it in range(STEPS):
    try:
        st = ops.train_step(ctx, P, w2c, Ks, campos, gt, w, h, 0.2, 0.01, 0.01, grads, m, vv, 1e-3, 0.9, 0.999, 1e-8, it + 1, losses[it:it + 1])
    except Exception as e:
        print("FAILED at", it, e, flush=True); break
    if it % 25 == 0 or it >= 195:
        torch.cuda.synchronize()
        print(it, st, "scale mean %.4f max %.3f" % (float(P["scales"].abs().mean()), float(P["scales"].abs().max())), "t %.1f" % (time.time()-t0), flush=True)
