"""Times the MCMC refinement hooks on SYNTH-1M-sized parameters (1 M Gaussians, 5 % dead, +5 % growth)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from starst3r_amd import ops

dev = torch.device("cuda:0"); ctx = ops.get_context(dev)
N, n_new = 1_000_000, 50_000
g = torch.Generator(device="cpu").manual_seed(0)
P = {"means": torch.randn(N + n_new, 3, generator=g), "quats": torch.randn(N + n_new, 4, generator=g),
     "scales": torch.randn(N + n_new, 3, generator=g) * 0.3 - 4, "opacities": torch.randn(N + n_new, generator=g) * 2,
     "sh0": torch.randn(N + n_new, 1, 3, generator=g), "shN": torch.randn(N + n_new, 24, 3, generator=g)}
P = {k: v.to(dev) for k, v in P.items()}
P["opacities"][:N][::20] = -8.0
m = torch.zeros(23 * N, device=dev); v = torch.zeros(23 * N, device=dev)
head = {k: t[:N] for k, t in P.items()}
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(3):
    P["opacities"][:N][::20] = -8.0
    torch.cuda.synchronize()
    e0 = ev(); ops.mcmc_relocate(ctx, head, m, v, 0.005, 1, it, want_count=False)
    e1 = ev(); ops.mcmc_add(ctx, P, N, n_new, 0.005, 1, it)
    e2 = ev(); ops.mcmc_noise(ctx, head, 500.0, 1, it)
    e3 = ev(); torch.cuda.synchronize()
    print(f"relocate {e0.elapsed_time(e1):.3f} ms  add {e1.elapsed_time(e2):.3f} ms  noise {e2.elapsed_time(e3):.3f} ms")
