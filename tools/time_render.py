"""Times st3r_gs_render at SYNTH-1M with the quadrant kernel and with the cell-list kernel (debug flag 512) and prints
the blend-forward stage time of both (HIP events of st3r_ctx_set_profiling are not wired into the render entry point,
so the whole call is timed: the front end is identical in both)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from starst3r_amd import ops
from st3r_synth import synth

N, V, W, H = 1_000_000, 8, 1920, 1080
g, w2c, Ks = synth.make_scene(N, V, W, H)
dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda:0")
P = {k: dev(v) for k, v in g.items()}
vm, K = dev(w2c), dev(Ks)
campos = ops.camera_positions(vm)
ctx = ops.get_context("cuda:0")
for flags in (0, 512, 0, 512):
    ops.set_debug(ctx, flags)
    for _ in range(3):
        ops.render(ctx, P, vm, K, campos, W, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.render(ctx, P, vm, K, campos, W, H)
    e1.record(); torch.cuda.synchronize()
    print(f"flags {flags}: render {e0.elapsed_time(e1) / 10:.3f} ms per call")
ops.set_debug(ctx, 0)
