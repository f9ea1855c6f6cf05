"""Phase times of k_isect_emit_chain (library built with ST3R_DEFS=-DEMIT_PROFILE): shader-clock cycles of thread 0 per
workgroup between ticket | gather | scan | look-back | emission.  python tools/emit_profile.py"""
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
gt = torch.rand(V, H, W, 3, device=dev)
grads = torch.empty(23 * N, device=dev); m = torch.zeros_like(grads); v = torch.zeros_like(grads); loss = torch.zeros(1, device=dev)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
for it in range(4):
    if it == 3:
        torch.cuda.synchronize(); L.st3r_debug_emit_profile(None, 1)
    ops.train_step(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 0.0, 0.9, 0.999, 1e-8, it + 1, loss)
torch.cuda.synchronize()
L.st3r_debug_emit_profile(buf, 0)
b = list(buf); n = max(b[7], 1)
print("workgroups", b[7], " cycles per workgroup (100 MHz realtime or shader clock, see ISA): ",
      {k: round(b[i] / n) for i, k in enumerate(["ticket", "gather", "scan", "lookback", "emit"])})
