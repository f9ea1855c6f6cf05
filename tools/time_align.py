import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from starst3r_amd import align
from st3r_synth import synth_align
for C in (2, 8, 32):
    flat = synth_align.flatten(synth_align.make_problem(n_views=C, n_corr=2000 // max(C - 1, 1) + 1, seed=1))
    align.run(flat, niter1=5, niter2=5); torch.cuda.synchronize()
    t0 = time.perf_counter(); res, par = align.run(flat); torch.cuda.synchronize(); t = time.perf_counter() - t0
    L = res["losses"].cpu().numpy()
    print(f"C={C} anchors={flat['anchor_idx'].size} corr={flat['corr_a1'].size} hip_total={t*1e3:.1f} ms loss {L[0]:.4f}->{L[499]:.4f} | {L[500]:.3f}->{L[-1]:.3f}")
