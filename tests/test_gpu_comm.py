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
    try:
        assert ops.get_exchange(ctx) == "allreduce"       # the default of a fresh communicator (ST3R_EXCHANGE unset)
        for mode in ("allreduce", "ranges", "rs_ag", "direct"):
            assert ops.set_exchange(ctx, mode) in ops.EXCHANGE_FORMS and ops.get_exchange(ctx) == mode
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
        ops.set_exchange(ctx, "allreduce")
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


def _one_step(ctx, P, w2c, Ks, gt, W, H, step=1, grads=None, m=None, v=None):
    from starst3r_amd import ops
    N = P["means"].shape[0]
    grads = torch.empty(23 * N, device=DEV) if grads is None else grads
    m = torch.zeros_like(grads) if m is None else m
    v = torch.zeros_like(grads) if v is None else v
    loss = torch.zeros(1, device=DEV)
    ops.train_step(ctx, P, w2c, Ks, ops.camera_positions(w2c), gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999,
                   1e-8, step, loss)
    torch.cuda.synchronize()
    return grads, m, v, loss


def test_range_exchange_of_a_chunked_call_equals_the_allreduce_result():
    """ranges + view chunks.  Round 3 staged only the FIRST chunk's gradients range-major, added the later
    chunks to the caller's buffer and then all-reduced that buffer alone: the first chunk's gradients were lost.  Every
    chunk now adds to the staged ranges and the rank takes part in the same K range collectives as an unchunked rank."""
    from starst3r_amd import dist as sdist, ops
    _, P, w2c, Ks, gt, W, H = _problem()
    ctx = ops.Context("cuda:0")          # a private context: the chunk count of a chunked call sticks to its context
    sdist.attach_native_comm(ctx)
    try:
        A = {k: t.clone() for k, t in P.items()}
        g_a, m_a, v_a, l_a = _one_step(ctx, A, w2c, Ks, gt, W, H)           # all-reduce, one pass
        ops.set_exchange(ctx, "ranges")
        ops.set_debug(ctx, 32)                                              # the training calls walk their views in two chunks
        B = {k: t.clone() for k, t in P.items()}
        g_b, m_b, v_b, l_b = _one_step(ctx, B, w2c, Ks, gt, W, H)
        ops.set_debug(ctx, 0)
        assert float(l_b) == pytest.approx(float(l_a), rel=1e-6)
        # chunked sums add the views in another order than the one-pass kernel: equal to rounding, not bit for bit
        scale = float(g_a.abs().max())
        assert float((g_b - g_a).abs().max()) <= 2e-6 * scale
        for k in A:
            assert torch.allclose(B[k], A[k], rtol=0, atol=2e-6), k          # one Adam step of lr 1e-3 on equal gradients
        assert torch.allclose(m_b, m_a, rtol=0, atol=2e-7 * scale)
    finally:
        sdist.detach_native_comm(ctx)
        ctx.close()


@pytest.mark.parametrize("form", ["allreduce", "ranges", "rs_ag", "direct"])
def test_a_failing_rank_still_takes_part_and_nobody_applies_the_step(form):
    """A step that fails on one rank must not strand the others in the collective: the failing rank issues
    every collective of the step and returns its error, the max-reduced status word skips the Adam update on the device
    everywhere, and the NEXT training call reports ST3R_ERR_PEER.  One rank here: it sees both its own error and the
    status word of the step."""
    from starst3r_amd import _lib, dist as sdist, ops
    ctx, P, w2c, Ks, gt, W, H = _problem()
    sdist.attach_native_comm(ctx)
    try:
        ops.set_exchange(ctx, form)
        A = {k: t.clone() for k, t in P.items()}
        N = A["means"].shape[0]
        grads = torch.zeros(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        ops.set_debug(ctx, 2048)                                            # this rank's forward/backward "fails"
        with pytest.raises(_lib.St3rError) as e1:
            _one_step(ctx, A, w2c, Ks, gt, W, H, 1, grads, m, v)
        assert e1.value.code == -4 and "simulated failure" in str(e1.value)
        ops.set_debug(ctx, 0)
        torch.cuda.synchronize()                                            # the collectives were issued and completed
        for k in A:
            assert torch.equal(A[k], P[k]), k                               # no update was applied
        assert not m.any() and not v.any()
        with pytest.raises(_lib.St3rError) as e2:                           # what every OTHER rank sees at its next call
            _one_step(ctx, A, w2c, Ks, gt, W, H, 1, grads, m, v)
        assert e2.value.code == -5
        for k in A:
            assert torch.equal(A[k], P[k]), k
        _one_step(ctx, A, w2c, Ks, gt, W, H, 1, grads, m, v)                # and the job continues
        ops.set_exchange(ctx, "allreduce")
        B = {k: t.clone() for k, t in P.items()}
        _one_step(ctx, B, w2c, Ks, gt, W, H, 1)
        for k in A:
            assert torch.equal(A[k], B[k]), (form, k)
        ops.settle(ctx)                                                     # nothing pending, nothing to report
    finally:
        ops.set_debug(ctx, 0)
        ops.set_exchange(ctx, "allreduce")
        sdist.detach_native_comm(ctx)


def test_standalone_adam_ignores_the_status_word_of_a_failed_exchanged_step():
    """the status word of a failed exchanged step stays in the counts buffer until the next exchanged step
    rewrites it.  Only st3r_gs_train_step's own update kernels may look at it: the documented composition
    st3r_gs_train_fwd_bwd -> st3r_grad_allreduce -> st3r_adam_step must apply its update under the communicator right
    after such a failure, and so must a single-process asynchronous step after the communicator is gone."""
    from starst3r_amd import _lib, dist as sdist, ops
    ctx, P, w2c, Ks, gt, W, H = _problem()
    N = P["means"].shape[0]
    campos = ops.camera_positions(w2c)
    sdist.attach_native_comm(ctx)
    attached = True
    try:
        A = {k: t.clone() for k, t in P.items()}
        grads = torch.zeros(23 * N, device=DEV); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        ops.set_debug(ctx, 2048)
        with pytest.raises(_lib.St3rError):
            _one_step(ctx, A, w2c, Ks, gt, W, H, 1, grads, m, v)           # leaves the status word at 1
        ops.set_debug(ctx, 0)
        loss = torch.zeros(1, device=DEV)
        # the two-call composition, still under the communicator, WITHOUT settling first: the stale word must not matter
        ops.train_fwd_bwd(ctx, A, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, loss, want_stats=True)
        _lib.check(_lib.lib().st3r_grad_allreduce(ctx.handle, ops._stream(), ops._p(grads), grads.numel()))
        ops.adam_step(ctx, A, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, 1)
        torch.cuda.synchronize()
        assert m.abs().max() > 0 and not torch.equal(A["means"], P["means"])   # applied, not silently skipped
        with pytest.raises(_lib.St3rError) as e:                            # the failed step is still reported, once
            ops.settle(ctx)
        assert e.value.code == -5
        B = {k: t.clone() for k, t in P.items()}
        _one_step(ctx, B, w2c, Ks, gt, W, H, 1)
        for k in A:
            assert torch.equal(A[k], B[k]), k
        # a second failure, then the communicator goes away: single-process steps (asynchronous ones included) apply
        ops.set_debug(ctx, 2048)
        with pytest.raises(_lib.St3rError):
            _one_step(ctx, A, w2c, Ks, gt, W, H, 2, grads, m, v)
        ops.set_debug(ctx, 0)
        sdist.detach_native_comm(ctx); attached = False
        before = A["means"].clone()
        for it in range(3):
            ops.train_step(ctx, A, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, 2 + it,
                           loss, want_stats=False)
            torch.cuda.synchronize()
            assert not torch.equal(A["means"], before), it
            before = A["means"].clone()
        ops.settle(ctx)
    finally:
        ops.set_debug(ctx, 0)
        if attached:
            sdist.detach_native_comm(ctx)


def test_allgather_pieces_replicates_piecewise_moments():
    from starst3r_amd import dist as sdist, ops
    ctx = ops.get_context(DEV)
    x = torch.arange(23 * 101, device=DEV, dtype=torch.float32); y = x.clone()
    ops.allgather_pieces(ctx, x)                       # no communicator: a no-op
    sdist.attach_native_comm(ctx)
    try:
        ops.allgather_pieces(ctx, x)                   # one rank: its piece is the whole buffer
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    finally:
        sdist.detach_native_comm(ctx)
