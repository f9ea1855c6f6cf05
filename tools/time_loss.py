"""Time the fused L1 + SSIM loss alone (8 views of 1920 x 1080, random images): tools/time_loss.py [reps]."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starst3r_amd import ops

ctx = ops.get_context("cuda:0")
torch.manual_seed(0)
x = torch.rand(8, 1080, 1920, 3, device="cuda:0"); y = torch.rand_like(x)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): ops.loss_l1_ssim(ctx, x, y, 0.8, 0.2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.loss_l1_ssim(ctx, x, y, 0.8, 0.2)
e1.record(); torch.cuda.synchronize()
print("loss_l1_ssim 8x1080p: %.4f ms per call (incl. the call's own memset / sums)" % (e0.elapsed_time(e1) / reps))
