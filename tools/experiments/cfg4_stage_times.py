"""Stage times of one rank's step of BASELINE configs[4] (5 M Gaussians, 8 views of 3840 x 2160) and of configs[3]'s rank workload
(300 k Gaussians, 4 views of 512 x 384): is any stage out of proportion with SYNTH-1M's?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from starst3r_amd import ops
from st3r_synth import synth
DEV = "cuda:0"
ctx = ops.get_context(DEV)
for name, (n, v, w, h) in {"cfg4 rank": (5_000_000, 8, 3840, 2160), "cfg3 rank": (300_000, 4, 512, 384)}.items():
    g, w2c_np, Ks_np = synth.make_scene(n, v, w, h, seed=21)
    P = {k: torch.tensor(val, device=DEV) for k, val in g.items()}
    P["shN"] = P["shN"][:, :4].contiguous()
    w2c = torch.tensor(w2c_np, device=DEV); Ks = torch.tensor(Ks_np, device=DEV)
    campos = ops.camera_positions(w2c)
    Q = {k: torch.tensor(val, device=DEV) for k, val in synth.perturb_for_gt(g).items()}
    gt, _, st0 = ops.render(ctx, Q, w2c, Ks, campos, w, h)
    gt = gt.clamp(0, 1).contiguous(); del Q
    mom = ops.gt_moments(ctx, gt); ops.set_gt_moments(ctx, gt, mom)
    grads = torch.empty(23 * n, device=DEV); m = torch.zeros_like(grads); vv = torch.zeros_like(grads)
    losses = torch.zeros(40, device=DEV)
    for it in range(40):
        if it == 10:
            ops.set_profiling(ctx, True); ops.stage_ms(ctx); torch.cuda.synchronize(); t0 = time.time()
        st = ops.train_step(ctx, P, w2c, Ks, campos, gt, w, h, 0.2, 0.01, 0.01, grads, m, vv, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                            losses[it:it + 1], want_stats=(it == 0))
        if it == 0:
            print(name, st, flush=True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 30 * 1e3
    stage = ops.stage_ms(ctx); ops.set_profiling(ctx, False); ops.set_gt_moments(ctx, None, None)
    print(name, "wall ms/step %.3f" % dt, {k: round(ms / c, 3) for k, (ms, c) in stage.items() if c}, flush=True)
    del P, gt, mom, grads, m, vv
    ops.release_scratch(); torch.cuda.empty_cache()
