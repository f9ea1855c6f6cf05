"""CPU restatement of the reference-OWNED glue of path C (TEST INFRASTRUCTURE ONLY -- nothing under starst3r_amd/
or in bench.py's timed region imports this).

  init_params   <- starster/gs.py:14-34   (`init_3dgs`: the six tensors)
  TrainLoop     <- starster/gs.py:37, 126-136, 143-161 (`compute_loss`, the loop body, one Adam per tensor)

Unlike the reference it does NOT go through autograd: the rasteriser is oracle/gs_oracle.c's forward + analytic
backward, the L1 + SSIM gradient comes from `gso_l1_ssim`, the two regularisers are differentiated by hand
(both are added once per VIEW, gs.py:150-152).  PINNED by tests/golden/gs_loop_*.npz, which tools/gen_gs_goldens.py
produces by executing /root/reference/starster/gs.py itself (autograd + torch.optim.Adam) over the same C rasteriser:
tests/test_oracle_gs_loop.py.  gsplat's / torchmetrics' own arithmetic stays [U] (parity unpinned vs upstream).
"""
import numpy as np
import torch

from . import gs_oracle as go

KEYS = ("means", "scales", "quats", "opacities", "sh0", "shN")
RENDERED = ("means", "quats", "scales", "opacities", "shN")   # sh0 is never rendered (gs.py:81), its grad stays None


def init_params(pts, cols, init_scale=3e-3):
    """gs.py:20-31: raw scales = init_scale, wxyz identity, raw opacity 1, `1 - colour` in sh0 and EVERY shN row."""
    pts = np.asarray(pts, np.float32); cols = np.asarray(cols, np.float32)
    n = pts.shape[0]
    g = dict(means=pts.copy(), scales=np.full_like(pts, init_scale), quats=np.zeros((n, 4), np.float32),
             opacities=np.ones(n, np.float32), sh0=np.zeros((n, 1, 3), np.float32), shN=np.zeros((n, 24, 3), np.float32))
    g["quats"][:, 0] = 1.0
    g["sh0"][:, 0] = 1 - cols
    g["shN"][:] = (1 - cols)[:, None, :]
    return g


class TrainLoop:
    """One `torch.optim.Adam([v], lr=lr)` per tensor (gs.py:37), stepped with hand-assembled gradients."""

    def __init__(self, params, lr=1e-3):
        self.P = {k: torch.tensor(np.asarray(params[k], np.float32), requires_grad=True) for k in KEYS}
        self.opt = {k: torch.optim.Adam([self.P[k]], lr=lr) for k in KEYS}

    def numpy(self):
        return {k: self.P[k].detach().numpy().copy() for k in KEYS}

    def step(self, imgs, w2c, Ks, W, H, ssim_fac=0.2, opac_fac=0.01, scale_fac=0.01):
        cur = {k: self.P[k].detach().numpy() for k in KEYS}
        N, V = cur["means"].shape[0], len(imgs)
        rgb, alpha, meta = go.rasterization(cur["means"], cur["quats"], cur["scales"], cur["opacities"], cur["shN"], w2c,
                                            Ks, W, H)
        v_rgb = np.zeros_like(rgb); loss = 0.0
        for c in range(V):                                    # gs.py:149-152, compute_loss :126-130
            l1, ss, vr = go.l1_ssim(rgb[c], imgs[c], 1 - ssim_fac, ssim_fac)
            loss += (1 - ssim_fac) * l1 + ssim_fac * (1 - ss); v_rgb[c] = vr
        G = go.rasterization_backward(cur["means"], cur["quats"], cur["scales"], cur["opacities"], cur["shN"], w2c, Ks,
                                      W, H, meta, alpha, v_rgb, None)
        sg = 1 / (1 + np.exp(-cur["opacities"].astype(np.float64))); ex = np.exp(cur["scales"].astype(np.float64))
        loss += V * (opac_fac * sg.mean() + scale_fac * ex.mean())        # gs.py:132,134 -- once per view
        g_op = G["opacities"] + V * opac_fac * sg * (1 - sg) / N
        g_sc = G["scales"] + V * scale_fac * ex / (3 * N)
        g_sh = np.zeros_like(cur["shN"]); g_sh[:, :4] = np.asarray(G["sh"]).reshape(N, 4, 3)   # rows 4..23: zeros
        for k, gk in (("means", G["means"]), ("quats", G["quats"]), ("scales", g_sc), ("opacities", g_op), ("shN", g_sh)):
            self.P[k].grad = torch.tensor(np.asarray(gk, np.float32).reshape(self.P[k].shape))
        for o in self.opt.values():                            # sh0: grad None -> Adam skips it, no state (gs.py:159-161)
            o.step(); o.zero_grad(set_to_none=True)
        return float(loss)
