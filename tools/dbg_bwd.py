import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gs_oracle as go
from starst3r_amd import ops
from st3r_synth import synth
ctx = ops.get_context("cuda:0")
N,V,W,H = 400,3,96,64
g,w2c,Ks = synth.make_scene(N,V,W,H,seed=7,scale_lo=0.01,scale_hi=0.08)
dev=lambda a: torch.tensor(a,dtype=torch.float32,device="cuda:0")
rgb_o,alpha_o,meta = go.rasterization(g["means"],g["quats"],g["scales"],g["opacities"],g["shN"],w2c,Ks,W,H)
rng=np.random.default_rng(3)
v_rgb=rng.standard_normal(rgb_o.shape).astype(np.float32)
v_alpha=rng.standard_normal(alpha_o.shape).astype(np.float32)
gg=go.rasterization_backward(g["means"],g["quats"],g["scales"],g["opacities"],g["shN"],w2c,Ks,W,H,meta,alpha_o,v_rgb,v_alpha)
P={k:dev(v) for k,v in g.items()}
rgb,alpha,info=ops.rasterization(ctx,P["means"],P["quats"],P["scales"],P["opacities"],P["shN"],dev(w2c),dev(Ks),W,H)
vs=ops.blend_bwd(ctx,info["_splats"],info["isect_offsets"],info["_flatten_ids_dense"],alpha,info["_last_ids"],dev(v_rgb),dev(v_alpha),info["_cum_tiles"],V,W,H)
torch.cuda.synchronize()
pid=(info["camera_ids"].long()*N+info["gaussian_ids"].long())
vs=vs[pid].cpu().numpy()
pk=gg["packed"]
ref=np.concatenate([pk["v_means2d"],pk["v_opacities"][:,None],pk["v_conics"],pk["v_colors"]],axis=1)
names=["x","y","o","ca","cb","cc","r","g","b"]
for k in range(9):
    errs=[np.abs(vs[:,k]-ref[:,j]).max()/(np.abs(ref[:,j]).max()+1e-20) for j in range(9)]
    print(names[k], "best match:", names[int(np.argmin(errs))], "err=%.2e"%min(errs), " own err=%.2e"%errs[k])
