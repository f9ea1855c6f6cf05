"""How loose are the tile rectangles of the configs[1] example's Gaussians?  For a sample of visible pairs after `warm` iterations:
tiles of the alpha >= 1/255 bounding box (what the fused path emits) against tiles the ELLIPSE itself reaches (exact test, the
arithmetic of tile_rect.h: ellipse_hits_square, evaluated in float64 torch)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import starst3r_amd as st
from starst3r_amd import ops
from st3r_synth.synth_model import SyntheticNetwork

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 512, 384
net = SyntheticNetwork(n_views=8, width=W, height=H, seed=2)
sc = st.Scene(device="cuda:0")
sc.add_images(net, net.images())
sc.init_3dgs()
ctx = ops.get_context("cuda:0")
sc.run_3dgs_optim(warm, enable_pruning=True)
g = sc.gaussians
N = g["means"].shape[0]
P = {k: g[k].data for k in ("means", "quats", "scales", "opacities", "shN")}
w2c = sc.w2c.to("cuda:0", torch.float32).contiguous(); Ks = sc.intrinsics.to("cuda:0", torch.float32).contiguous()
gt = torch.stack([torch.as_tensor(np.asarray(i), dtype=torch.float32) for i in sc.imgs]).cuda().contiguous()
grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
stt = ops.train_fwd_bwd(ctx, P, w2c, Ks, ops.camera_positions(w2c), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
print("stats", stt)
S = ops.peek(ctx, 2, 8 * N * 12, torch.float32).reshape(-1, 12).double()
vis = S[:, 10].float().view(torch.int32) > 0
S = S[vis]
idx = torch.randperm(S.shape[0], device=S.device)[:200000]
S = S[idx]
x, y, o, A, B, C = S[:, 0], S[:, 1], S[:, 2], S[:, 3], S[:, 4], S[:, 5]
ok = o * 255 > 1
tau = torch.log((o * 255).clamp(min=1.0000001))
det = A * C - B * B
ex = torch.sqrt(2 * tau * C / det); ey = torch.sqrt(2 * tau * A / det)
tw, th = W // 16, H // 16
tx0 = torch.ceil((x - ex - 15.5) / 16).clamp(0, tw); tx1 = (torch.floor((x + ex - 0.5) / 16) + 1).clamp(0, tw)
ty0 = torch.ceil((y - ey - 15.5) / 16).clamp(0, th); ty1 = (torch.floor((y + ey - 0.5) / 16) + 1).clamp(0, th)
w_ = (tx1 - tx0).clamp(min=0); h_ = (ty1 - ty0).clamp(min=0)
area = (w_ * h_) * ok
K = 32
jx = torch.arange(K, device=S.device)[None, :, None].double(); jy = torch.arange(K, device=S.device)[None, None, :].double()
inr = (jx < w_[:, None, None]) & (jy < h_[:, None, None]) & ok[:, None, None]
dx0 = (tx0[:, None, None] + jx) * 16 + 0.5 - x[:, None, None]; dx1 = dx0 + 15
dy0 = (ty0[:, None, None] + jy) * 16 + 0.5 - y[:, None, None]; dy1 = dy0 + 15
Ae, Be, Ce, te = A[:, None, None], B[:, None, None], C[:, None, None], tau[:, None, None]
hit = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
sig = lambda dx, dy: 0.5 * (Ae * dx * dx + Ce * dy * dy) + Be * dx * dy
for dxe in (dx0, dx1):
    dyc = torch.minimum(torch.maximum(-Be / Ce * dxe, dy0), dy1)
    hit |= sig(dxe, dyc) <= te
for dye in (dy0, dy1):
    dxc = torch.minimum(torch.maximum(-Be / Ae * dye, dx0), dx1)
    hit |= sig(dxc, dye) <= te
exact = (hit & inr).sum((1, 2)).double()
big = (w_ <= K) & (h_ <= K)
print("sampled visible pairs", S.shape[0], "mean rectangle tiles %.2f" % float(area[big].mean()), "mean exact tiles %.2f" % float(exact[big].mean()),
      "kept fraction %.3f" % float(exact[big].sum() / area[big].sum()))
for lo, hi in ((1, 4), (5, 16), (17, 64), (65, 1024)):
    m = big & (area >= lo) & (area <= hi)
    if m.any():
        print(f"  rectangles of {lo}..{hi} tiles: {float(m.float().mean()):.3f} of the pairs, {float(area[m].sum() / area[big].sum()):.3f} of the records, kept {float(exact[m].sum() / area[m].sum()):.3f}")
