"""Where does an iteration of the configs[1] example (tests/test_gpu_configs.py: 8 views of 512 x 384, ~0.9 M Gaussians seeded from
the dense points) spend its time?  Stage times (HIP events) after `warm` iterations, and the wall time per iteration."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import starst3r_amd as st
from starst3r_amd import ops
from st3r_synth.synth_model import SyntheticNetwork

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
net = SyntheticNetwork(n_views=8, width=512, height=384, seed=2)
sc = st.Scene(device="cuda:0")
sc.add_images(net, net.images())
sc.init_3dgs()
ctx = ops.get_context("cuda:0")
sc.run_3dgs_optim(warm, enable_pruning=True)
for pruning in (True, False):
    ops.set_profiling(ctx, True); ops.stage_ms(ctx)
    torch.cuda.synchronize(); t0 = time.time()
    sc.run_3dgs_optim(50, enable_pruning=pruning)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 50 * 1e3
    stage = ops.stage_ms(ctx); ops.set_profiling(ctx, False)
    print("pruning", pruning, "N", sc.gaussians["means"].shape[0], "wall ms/iter %.3f" % dt,
          {k: round(ms / n, 3) for k, (ms, n) in stage.items() if n}, "sum %.3f" % sum(ms / n for ms, n in stage.values() if n))
