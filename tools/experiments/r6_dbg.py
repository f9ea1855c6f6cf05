import numpy as np, torch, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from starst3r_amd import ops
import test_gpu_gs as T
ctx = ops.get_context("cuda:0")
dev = T.dev
g, w2c, Ks, W, H = T.make("small")
N, Cn = g["means"].shape[0], w2c.shape[0]
# a normal step first (leaves a non-black image in the scratch)
P = {k: dev(v) for k, v in g.items()}
vm, K = dev(w2c), dev(Ks)
gt = torch.rand((Cn, H, W, 3), device="cuda:0")
grads = torch.empty(23 * N, device="cuda:0"); loss = torch.zeros(1, device="cuda:0")
st = ops.train_fwd_bwd(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
torch.cuda.synchronize(); print("normal", st, float(ops.peek(ctx, 8, Cn * H * W * 3, torch.float32).abs().max()))
far = {k: v.copy() for k, v in g.items()}
far["means"] = (far["means"] * 0.01 + np.array([100.0, 0, 0], np.float32)).astype(np.float32)
P = {k: dev(v) for k, v in far.items()}
st = ops.train_fwd_bwd(ctx, P, vm, K, ops.camera_positions(vm), gt, W, H, 0.2, 0.01, 0.01, grads, loss)
torch.cuda.synchronize()
print("far", st, float(loss))
tw, th = ops.tile_grid(W, H)
off = ops.peek(ctx, 1, Cn * tw * th + 1)
print("offsets", off[:10].tolist(), int(off.abs().max()))
print("rgb max", float(ops.peek(ctx, 8, Cn * H * W * 3, torch.float32).abs().max()))
print("alpha max", float(ops.peek(ctx, 9, Cn * H * W, torch.float32).abs().max()))
