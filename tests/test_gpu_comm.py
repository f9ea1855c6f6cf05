"""The native exchange step on a single GPU: the library's own RCCL communicator (world size 1 -- the
multi-rank semantics are covered by tests/test_dist_cpu.py on gloo), the in-place gradient all-reduce, and
st3r_gs_train_step == st3r_gs_train_fwd_bwd + st3r_adam_step."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _problem(N=6000, V=2, W=96, H=64):   # >= 4096 Gaussians: the range-wise exchange splits them
    from starst3r_amd import ops
    from st3r_synth import synth
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
    # one-call path, the exchange inside -- every form of it (csrc/comm.hip: plain all-reduce; range-wise all-reduce on a
    # second stream under the projection backward / Adam of the neighbouring ranges; reduce-scatter -> Adam on the own
    # piece -> parameter all-gather).  One rank: the collectives are copies, the control flow, streams, events, index
    # ranges and the staging buffer are the real ones.
    import os
    try:
        for mode in ("allreduce", "ranges", "rs_ag"):
            os.environ["ST3R_EXCHANGE"] = mode
            B = {k: v_.clone() for k, v_ in P.items()}
            grads_b = torch.empty(23 * N, device=DEV); mb = torch.zeros_like(grads); vb = torch.zeros_like(grads)
            loss_b = torch.zeros(1, device=DEV)
            ops.train_step(ctx, B, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads_b, mb, vb, 1e-3, 0.9, 0.999, 1e-8, 1, loss_b)
            torch.cuda.synchronize()
            assert float(loss_a) == pytest.approx(float(loss_b), rel=1e-6), mode
            # same kernels, same sums: the gradients, moments and parameters of the two-call path, bit for bit
            assert torch.equal(grads_b, grads), mode
            assert torch.equal(mb, m) and torch.equal(vb, v), mode
            for k in A:
                assert torch.equal(B[k], A[k]), (mode, k)
    finally:
        os.environ.pop("ST3R_EXCHANGE", None)
        sdist.detach_native_comm(ctx)
    assert not ctx.native_comm


def test_sharded_adam_pieces_reassemble_the_replicated_update():
    """The reduce-scatter exchange (ST3R_EXCHANGE=rs_ag) on one GPU with the collectives done by hand: w virtual ranks,
    each with its own copy of the parameters and moments, run Adam on their piece [r q, (r+1) q) of the (already summed)
    gradient buffer (st3r_adam_step_range), the pieces of the staging buffers are "all-gathered" with a copy,
    st3r_params_from_stage fills the rest, everyone updates the < w floats of the tail -- every rank then holds exactly
    the parameters of the replicated st3r_adam_step, and its piece of the moments."""
    from starst3r_amd import ops
    ctx = ops.get_context(DEV)
    N, w = 1003, 4                       # 23 N = 23069 = 4 x 5767 + 1: a tail of one float
    torch.manual_seed(5)
    P0 = dict(means=torch.randn(N, 3, device=DEV), quats=torch.randn(N, 4, device=DEV), scales=torch.randn(N, 3, device=DEV),
              opacities=torch.randn(N, device=DEV), shN=torch.randn(N, 24, 3, device=DEV))
    grads = torch.randn(23 * N, device=DEV)
    m0 = 0.1 * torch.randn(23 * N, device=DEV); v0 = 0.01 * torch.rand(23 * N, device=DEV)
    ref = {k: t.clone() for k, t in P0.items()}; m_ref, v_ref = m0.clone(), v0.clone()
    ops.adam_step(ctx, ref, grads, m_ref, v_ref, 1e-3, 0.9, 0.999, 1e-8, 7)
    total = 23 * N; q = total // w; tail0 = q * w
    ranks = [dict(P={k: t.clone() for k, t in P0.items()}, m=m0.clone(), v=v0.clone(), stage=torch.zeros(total, device=DEV))
             for _ in range(w)]
    for r, R in enumerate(ranks):
        ops.adam_step_range(ctx, R["P"], grads, R["m"], R["v"], 1e-3, 0.9, 0.999, 1e-8, 7, r * q, (r + 1) * q, R["stage"])
        ops.adam_step_range(ctx, R["P"], grads, R["m"], R["v"], 1e-3, 0.9, 0.999, 1e-8, 7, tail0, total)
    gathered = torch.zeros(total, device=DEV)
    for r, R in enumerate(ranks):
        gathered[r * q:(r + 1) * q] = R["stage"][r * q:(r + 1) * q]
    for r, R in enumerate(ranks):
        ops.params_from_stage(ctx, R["P"], gathered, r * q, (r + 1) * q, tail0)
    torch.cuda.synchronize()
    for r, R in enumerate(ranks):
        for k in ref:
            if k == "shN":   # rows 4..23 are never touched
                assert torch.equal(R["P"][k][:, 4:], P0[k][:, 4:])
            assert torch.equal(R["P"][k], ref[k]), (r, k)
        assert torch.equal(R["m"][r * q:(r + 1) * q], m_ref[r * q:(r + 1) * q])
        assert torch.equal(R["v"][tail0:], v_ref[tail0:])
        other = (r + 1) % w
        assert torch.equal(R["m"][other * q:(other + 1) * q], m0[other * q:(other + 1) * q])   # not this rank's piece
