"""End to end on synthetic data (the role of the reference's main.py, without the Mast3r weights): a stand-in network
supplies per-pair head outputs; pair list, reciprocal matching, pair cache, condensation, global alignment, dense
seeding and 3DGS refinement with the MCMC hooks run in the library.

    python examples/synthetic_end_to_end.py [views] [iterations]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import starst3r_amd as st
from st3r_synth.synth_model import SyntheticNetwork


def psnr(sc, W, H):
    vals = []
    for v in range(len(sc.imgs)):
        img, _, _ = sc.render_3dgs(torch.inverse(sc.c2w[v])[None], sc.intrinsics[v][None], W, H)
        gt = torch.as_tensor(np.asarray(sc.imgs[v]), device=img.device).reshape(H, W, 3)
        vals.append(-10 * np.log10(float(((img[0].detach().clamp(0, 1) - gt) ** 2).mean())))
    return float(np.mean(vals))


def main(views=4, iters=1200, W=256, H=192):
    net = SyntheticNetwork(n_views=views, width=W, height=H, seed=2)
    sc = st.Scene(device="cuda:0")
    t0 = time.time()
    sc.add_images(net, net.images())
    torch.cuda.synchronize()
    print(f"reconstruction of {views} views: {time.time() - t0:.2f} s ({net.calls} pair inferences)")
    sc.init_3dgs()
    before = psnr(sc, W, H)
    t0 = time.time()
    sc.run_3dgs_optim(iters, enable_pruning=True, verbose=False)
    torch.cuda.synchronize()
    after = psnr(sc, W, H)
    print(f"{iters} iterations: {time.time() - t0:.2f} s, {sc.gaussians['means'].shape[0]} Gaussians, "
          f"PSNR {before:.1f} -> {after:.1f} dB")
    return before, after


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    main(*a)
