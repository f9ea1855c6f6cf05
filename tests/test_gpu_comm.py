"""The native exchange step on a single GPU: the library's own RCCL communicator (world size 1 -- the
multi-rank semantics are covered by tests/test_dist_cpu.py on gloo), the in-place gradient all-reduce, and
st3r_gs_train_step == st3r_gs_train_fwd_bwd + st3r_adam_step."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _problem(N=4000, V=2, W=96, H=64):
    from starst3r_amd import ops, synth
    ctx = ops.get_context(DEV)
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=5, scale_lo=0.01, scale_hi=0.05)
    P = {k: torch.from_numpy(g[k]).to(DEV) for k in ("means", "quats", "scales", "opacities", "shN")}
    w2c = torch.from_numpy(w2c).to(DEV); Ks = torch.from_numpy(Ks).to(DEV)
    gt_g = synth.perturb_for_gt(g, sigma=0.01)
    Q = {k: torch.from_numpy(gt_g[k]).to(DEV) for k in P}
    gt, _, _ = ops.render(ctx, Q, w2c, Ks, ops.camera_positions(w2c), W, H)
    return ctx, P, w2c, Ks, gt.clamp(0, 1).contiguous(), W, H


def test_native_comm_single_rank_and_fused_step():
    from starst3r_amd import _lib, dist as sdist, ops
    ctx, P, w2c, Ks, gt, W, H = _problem()
    N = P["means"].shape[0]
    campos = ops.camera_positions(w2c)
    # two-call path
    A = {k: v.clone() for k, v in P.items()}
    grads = torch.empty(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    loss_a = torch.zeros(1, device=DEV)
    ops.train_fwd_bwd(ctx, A, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss_a)
    ops.adam_step(ctx, A, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, 1)
    # communicator owned by the library
    assert sdist.attach_native_comm(ctx) == (0, 1) and ctx.native_comm
    ws, rk = C.c_int(-1), C.c_int(-1)
    _lib.check(_lib.lib().st3r_comm_world(ctx.handle, C.byref(ws), C.byref(rk)))
    assert (ws.value, rk.value) == (1, 0)
    with pytest.raises(ValueError):          # a second communicator on the same ctx is refused
        sdist.attach_native_comm(ctx)
    x = torch.arange(1000, device=DEV, dtype=torch.float32); y = x.clone()
    _lib.check(_lib.lib().st3r_grad_allreduce(ctx.handle, ops._stream(), ops._p(x), x.numel()))
    torch.cuda.synchronize()
    assert torch.equal(x, y)                 # sum over one rank
    # one-call path, all-reduce inside
    B = {k: v_.clone() for k, v_ in P.items()}
    grads_b = torch.empty(23 * N, device=DEV); mb = torch.zeros_like(grads); vb = torch.zeros_like(grads)
    loss_b = torch.zeros(1, device=DEV)
    ops.train_step(ctx, B, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads_b, mb, vb, 1e-3, 0.9, 0.999, 1e-8, 1, loss_b)
    sdist.detach_native_comm(ctx)
    assert not ctx.native_comm
    assert float(loss_a) == pytest.approx(float(loss_b), rel=1e-6)
    np.testing.assert_allclose(grads_b.cpu().numpy(), grads.cpu().numpy(), rtol=1e-4, atol=1e-7)
    for k in A:
        # Adam's first step moves every touched parameter by ~lr * sign(g): compare with that scale
        np.testing.assert_allclose(B[k].cpu().numpy(), A[k].cpu().numpy(), rtol=0, atol=2.1e-3 if k != "shN" else 2.1e-3)
        same = (B[k] == A[k]).float().mean().item()
        assert same > 0.99, (k, same)
