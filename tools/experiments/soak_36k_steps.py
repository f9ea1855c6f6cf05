import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from starst3r_amd import ops
from st3r_synth import synth
dev = "cuda:0"
ctx = ops.Context(dev)
N, V, W, H = 30000, 3, 320, 240
g, w2c, Ks = synth.make_scene(N, V, W, H, seed=5, scale_lo=0.004, scale_hi=0.03)
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
vm, K = T(w2c), T(Ks); campos = ops.camera_positions(vm)
P = {k: T(v) for k, v in g.items()}
Q = {k: T(v) for k, v in synth.perturb_for_gt(g).items()}
gt, _, _ = ops.render(ctx, Q, vm, K, campos, W, H); gt = gt.clamp(0, 1).contiguous()
grads = torch.empty(23 * N, device=dev); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
steps = 36000
losses = torch.zeros(steps, device=dev)
t0 = time.time()
mom = ops.gt_moments(ctx, gt); ops.set_gt_moments(ctx, gt, mom)   # (round 6: like gs.run_3dgs_optim)
passes = set()
for it in range(steps):
    if it % 500 == 0:
        passes.add(int(ops.peek(ctx, 10, 16)[10]))                # pass count of the segmented level-1 sort so far
    ops.train_step(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1, losses[it:it+1], want_stats=(it == 0))
ops.settle(ctx)
torch.cuda.synchronize()
L = losses.cpu().numpy()
print("level-1 passes seen", sorted(passes)); print("steps", steps, "sec", time.time() - t0, "finite", np.isfinite(L).all(), "loss", L[0], L[1000], L[16380:16390], L[32760:32775], L[-1])
for k, t in P.items(): assert torch.isfinite(t).all(), k
