"""run-to-run reproducibility of the synthetic end-to-end pipeline: reconstruct twice, compare poses/points bit for bit"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import starst3r_amd as st
from st3r_synth.synth_model import SyntheticNetwork
outs = []
for rep in range(3):
    net = SyntheticNetwork(n_views=3, width=128, height=96, seed=2)
    sc = st.Scene(device="cuda:0")
    sc.add_images(net, net.images())
    torch.cuda.synchronize()
    c2w = sc.c2w.detach().cpu().numpy().copy(); K = sc.intrinsics.detach().cpu().numpy().copy()
    n = sc.dense_pts_flat.shape[0]
    outs.append((c2w, K, n, sc.dense_pts_flat.double().sum().item()))
    print(rep, "n pts", n, "K00", K[:, 0, 0], "c2w[1] t", c2w[1, :3, 3])
for a, b in zip(outs[:-1], outs[1:]):
    print("identical c2w", np.array_equal(a[0], b[0]), "K", np.array_equal(a[1], b[1]), "n", a[2] == b[2], "max |dc2w|", np.abs(a[0] - b[0]).max())
