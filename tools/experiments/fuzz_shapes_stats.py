import sys; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, torch
from test_gpu_gs import _fuzz_shapes, dev
from starst3r_amd import ops
from st3r_synth import synth
ctx = ops.get_context("cuda:0")
for (N,V,W,H,lo,hi,seed) in _fuzz_shapes():
    g,w2c,Ks = synth.make_scene(N,V,W,H,seed=seed,scale_lo=lo,scale_hi=hi)
    P={k:dev(v) for k,v in g.items()}; vm,K=dev(w2c),dev(Ks)
    rgb,alpha,info=ops.rasterization(ctx,P["means"],P["quats"],P["scales"],P["opacities"],P["shN"],vm,K,W,H)
    print(N,V,W,H,lo,hi,"isects",info["isect_ids"].numel())
