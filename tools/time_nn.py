import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from starst3r_amd import matching, ops
ctx = ops.get_context("cuda:0")
H, W, D = 384, 512, 24
g = torch.Generator(device="cuda:0").manual_seed(0)
A = torch.nn.functional.normalize(torch.randn(H * W, D, device="cuda:0", generator=g), dim=1)
B = torch.nn.functional.normalize(torch.randn(H * W, D, device="cuda:0", generator=g), dim=1)
for n in (3072, 1024, 256):
    q = A[:n].contiguous()
    matching.nn_dot_argmax(ctx, q, B); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): matching.nn_dot_argmax(ctx, q, B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"n={n} m={H*W}: {ms*1e3:.1f} us  {2*n*H*W*D/ms/1e9:.1f} TFLOP/s (fp32 MFMA peak 157)")
t0 = time.perf_counter()
i1, i2 = matching.fast_reciprocal_NNs(A.reshape(H, W, D), B.reshape(H, W, D), 8, ret_xy=False, device="cuda:0"); torch.cuda.synchronize()
print("fast_reciprocal_NNs 512x384 subsample 8:", (time.perf_counter() - t0) * 1e3, "ms, matches:", i1.numel())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn_oracle as no
A2, B2, _, _ = no.synth_descriptors(H, W, planted=0.3, seed=1)
A2 = torch.from_numpy(A2).cuda(); B2 = torch.from_numpy(B2).cuda()
matching.fast_reciprocal_NNs(A2, B2, 8, ret_xy=False, device="cuda:0"); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): i1, i2 = matching.fast_reciprocal_NNs(A2, B2, 8, ret_xy=False, device="cuda:0")
torch.cuda.synchronize()
print("device-resident fast_reciprocal_NNs 512x384 (30% planted):", (time.perf_counter() - t0) / 5 * 1e3, "ms, matches:", i1.numel())
