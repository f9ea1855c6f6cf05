"""View-sharded data parallelism for the 3DGS train step.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).
The loss is a plain sum over views (starster/gs.py:149-152), so dL/dtheta = sum over views: every rank
renders its own views, the [23N] gradient buffer is sum-all-reduced once per iteration, and every rank
applies the identical fused Adam update to its replica of the Gaussians.  The two regularisers are added
once per view in the reference (gs.py:150-152): each rank therefore adds them C_local times and the
all-reduce restores the factor C.  The reference itself is single-device; nothing here mirrors a
reference call pattern.
"""
import torch


def rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_views(n_views, rank, world):
    """Round-robin view assignment; every rank must own at least one view."""
    views = list(range(rank, n_views, world))
    if not views:
        raise ValueError(f"rank {rank} of {world} got no view out of {n_views}: use at most n_views ranks")
    return views


def all_reduce_sum(t):
    """In-place sum over ranks (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() != "nccl":   # a host-side process group (gloo): staged through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_varlen(t):
    """Variable-length all-gather ("allgatherv") of a 1-D tensor: returns the list of every rank's tensor, in rank
    order.  Sizes are exchanged first, the payload travels padded to the longest (one collective, no per-rank
    point-to-point schedule: inside a node every GPU pair has its own xGMI link).  Single process: [t]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [t]
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=t.dtype, device=t.device)
    send[:t.numel()] = t.reshape(-1)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    return [r[:k] for r, k in zip(recv, sizes)]


def shard_pairs(n_pairs, rank, world):
    """Path A: the image pairs of the complete graph (starster/reconstruct.py:52) are independent --
    round-robin over the ranks, no collective on the data path."""
    return list(range(rank, n_pairs, world))


def sharded_step(local_views_fn, grads, loss):
    """One data-parallel step given a function that fills `grads` / `loss` from this rank's views."""
    local_views_fn(grads, loss)
    all_reduce_sum(grads)
    all_reduce_sum(loss)
    return grads, loss


def attach_native_comm(ctx):
    """Create the library's own RCCL communicator for `ctx` (st3r_comm_init) so that the whole iteration --
    render, loss, backward, gradient all-reduce, Adam -- is one C call (st3r_gs_train_step) with no Python
    between the kernels and the collective.  torch.distributed only ships the 128-byte id from rank 0.
    Collective: every rank must call it.  Returns (rank, world)."""
    import ctypes as C
    import torch.distributed as dist
    from . import _lib
    rank, world = rank_world()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.check(_lib.lib().st3r_comm_unique_id(buf))
    if world > 1:
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0)
        buf = C.create_string_buffer(box[0], 128)
    _lib.check(_lib.lib().st3r_comm_init(ctx.handle, world, rank, buf))
    ctx.native_comm = True
    return rank, world


def detach_native_comm(ctx):
    from . import _lib
    _lib.check(_lib.lib().st3r_comm_destroy(ctx.handle))
    ctx.native_comm = False


# ----------------------------------------------------------------------------------------------------------
# Gaussian-sharded mode (include/st3r.h, "Gaussian-sharded multi-GPU mode"): rank r owns Gaussians
# [r N/w, (r+1) N/w) and views [r C, (r+1) C).  Two all-to-alls per iteration move splat records to the views'
# owners and their gradients back; nothing is replicated and nothing is all-reduced except the reported loss.
# ----------------------------------------------------------------------------------------------------------

def shard_gaussians(N, rank, world):
    """Rows [lo, hi) of rank `rank`: contiguous, balanced (sizes differ by at most one when world does not divide N --
    the MCMC growth steps do not keep N divisible)."""
    return N * rank // world, N * (rank + 1) // world


def shard_counts(N, world):
    return [N * (r + 1) // world - N * r // world for r in range(world)]


def shard_views_contiguous(n_views, rank, world):
    if n_views % world:
        raise ValueError(f"the Gaussian-sharded mode needs the views ({n_views}) divisible by the ranks ({world})")
    c = n_views // world
    return list(range(rank * c, (rank + 1) * c))


def _all_to_all(recv, send, recv_splits=None, send_splits=None):
    """all_to_all_single; the split lists (elements per peer) are None for equal chunks."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():   # also with a single rank: the same RCCL call path
        dist.all_to_all_single(recv, send, recv_splits, send_splits)
    else:
        recv.copy_(send)


def records_to_view_owners(records_local, world, a2a=_all_to_all, recv=None, out=None, counts=None):
    """records_local [V, n, K] (own Gaussians, all V = world*C views) -> [C, N, K]: all Gaussians (global order)
    for the C views this rank owns.  Chunk r of the send buffer = views of rank r.  recv / out: optional receive
    buffer and (for C > 1) the buffer the view-major result is written to -- with both given nothing is allocated.
    counts: Gaussians per rank when the shards are uneven (None: every rank holds n)."""
    V, n, K = records_local.shape
    C = V // world
    even = counts is None or all(c == n for c in counts)
    send = records_local.reshape(world, C, n, K)
    if even:
        recv = torch.empty_like(send) if recv is None else recv.reshape(world, C, n, K)   # [source rank, C, n, K]
        a2a(recv.reshape(-1), send.reshape(-1))
        if C == 1:
            return recv.reshape(1, world * n, K)
        if out is None:
            out = torch.empty((C, world * n, K), dtype=recv.dtype, device=recv.device)
        out.reshape(C, world, n, K).copy_(recv.permute(1, 0, 2, 3))
        return out.reshape(C, world * n, K)
    N = sum(counts)
    recv = torch.empty(C * N * K, dtype=send.dtype, device=send.device) if recv is None else recv.reshape(-1)
    a2a(recv, send.reshape(-1), [C * c * K for c in counts], [C * n * K] * world)   # source s sends [C, n_s, K]
    if C == 1:
        return recv.reshape(1, N, K)
    if out is None:
        out = torch.empty((C, N, K), dtype=recv.dtype, device=recv.device)
    out = out.reshape(C, N, K)
    off = 0
    for c in counts:
        out[:, off:off + c].copy_(recv[C * off * K:C * (off + c) * K].reshape(C, c, K))
        off += c
    return out


def records_to_gaussian_owners(v_records, world, a2a=_all_to_all, recv=None, send_buf=None, counts=None, rank=0):
    """v_records [C, N, K] (own views, all Gaussians) -> [V, n, K]: all views for the own Gaussians.
    recv / send_buf: optional preallocated buffers (send_buf only matters for C > 1).  counts / rank: uneven shards."""
    C, N, K = v_records.shape
    even = counts is None or all(c == counts[0] for c in counts)
    if even:
        n = N // world
        send = v_records.reshape(C, world, n, K)
        if C == 1:
            send = send.reshape(world, n, K)
        else:                                                                              # [dest rank, C, n, K]
            if send_buf is None:
                send_buf = torch.empty((world, C, n, K), dtype=v_records.dtype, device=v_records.device)
            send_buf = send_buf.reshape(world, C, n, K)
            send_buf.copy_(send.permute(1, 0, 2, 3))
            send = send_buf
        if recv is None:
            recv = torch.empty((world, C, n, K), dtype=v_records.dtype, device=v_records.device)
        recv = recv.reshape(world, C, n, K)                                                # [source = view owner, C, n, K]
        a2a(recv.reshape(-1), send.reshape(-1))
        return recv.reshape(world * C, n, K)
    n = counts[rank]
    if C == 1:
        send = v_records.reshape(-1)              # the owners' row ranges are already contiguous and in rank order
    else:
        if send_buf is None:
            send_buf = torch.empty(C * N * K, dtype=v_records.dtype, device=v_records.device)
        send = send_buf.reshape(-1)
        off = 0
        for c in counts:
            send[C * off * K:C * (off + c) * K].reshape(C, c, K).copy_(v_records[:, off:off + c])
            off += c
    if recv is None:
        recv = torch.empty((world, C, n, K), dtype=v_records.dtype, device=v_records.device)
    recv = recv.reshape(-1)[:world * C * n * K]
    a2a(recv, send, [C * n * K] * world, [C * c * K for c in counts])
    return recv.reshape(world * C, n, K)


class ShardedTrainer:
    """One rank of the Gaussian-sharded train loop.  `params` hold only this rank's Gaussians
    (means/quats/scales/opacities/shN rows [lo, hi)); w2c/Ks describe ALL V views; gt holds the images of the
    views this rank owns (shard_views_contiguous)."""

    def __init__(self, ctx, params, n_total, w2c_all, Ks_all, gt_local, W, H, rank, world, lr=1e-3,
                 ssim_fac=0.2, opac_fac=0.01, scale_fac=0.01, a2a=_all_to_all, counts=None):
        from . import ops
        self.ops, self.ctx, self.P, self.N, self.W, self.H = ops, ctx, params, n_total, W, H
        self.rank, self.world, self.a2a = rank, world, a2a
        self.w2c, self.Ks, self.gt = w2c_all.contiguous(), Ks_all.contiguous(), gt_local
        self.campos = ops.camera_positions(self.w2c)
        self.V = self.w2c.shape[0]
        self.C = self.V // world
        self.n = params["means"].shape[0]
        self.counts = list(counts) if counts is not None else shard_counts(n_total, world)
        assert sum(self.counts) == n_total and self.counts[rank] == self.n, "shard sizes do not add up to n_total"
        assert self.C * world == self.V and gt_local.shape[0] == self.C
        dev = params["means"].device
        self.grads = torch.empty(23 * self.n, device=dev)
        self.m = torch.zeros_like(self.grads); self.v = torch.zeros_like(self.grads)
        self.v_records = torch.empty((self.C * n_total, 12), device=dev)
        # steady-state buffers: nothing is allocated inside step()
        self.rec = torch.empty((self.V * self.n, 12), device=dev)
        self.tiles = torch.empty((self.V * self.n,), dtype=torch.int32, device=dev)
        self.recv_fwd = torch.empty((self.C * n_total, 12), device=dev)
        self.recv_bwd = torch.empty((self.V * self.n, 12), device=dev)
        # view-major / rank-major copies of the exchanged records (only needed with several views per rank)
        self.mine_buf = torch.empty((self.C * n_total, 12), device=dev) if self.C > 1 else None
        self.send_bwd = torch.empty((self.C * n_total, 12), device=dev) if self.C > 1 else None
        self.reg = torch.zeros(4, dtype=torch.float64, device=dev)
        self.kreg = torch.tensor([self.V * opac_fac / n_total, self.V * scale_fac / (3 * n_total)], dtype=torch.float64,
                                 device=dev)
        self.lr, self.ssim_fac, self.opac_fac, self.scale_fac, self.t = lr, ssim_fac, opac_fac, scale_fac, 0

    def step(self, loss_out):
        """loss_out[0] receives this rank's part of the loss (sum over ranks = the reference's loss)."""
        ops, P = self.ops, self.P
        self.reg.zero_()
        rec, _ = ops.project_sh(self.ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], self.w2c,
                                self.Ks, self.campos, self.W, self.H, reg_sums=self.reg, out=(self.rec, self.tiles))
        mine = records_to_view_owners(rec.reshape(self.V, self.n, 12), self.world, self.a2a, recv=self.recv_fwd,
                                      out=self.mine_buf, counts=self.counts)
        st = ops.raster_train(self.ctx, mine.reshape(-1, 12), self.N, self.C, self.gt, self.W, self.H, self.ssim_fac,
                              self.v_records, loss_out)
        back = records_to_gaussian_owners(self.v_records.reshape(self.C, self.N, 12), self.world, self.a2a,
                                          recv=self.recv_bwd, send_buf=self.send_bwd, counts=self.counts, rank=self.rank)
        frac = self.n / self.N    # the regularisers are means over ALL N Gaussians (starster/gs.py:132,134)
        ops.project_sh_bwd(self.ctx, P["means"], P["quats"], P["scales"], P["opacities"], P["shN"], self.w2c,
                           self.Ks, self.campos, self.W, self.H, rec, back.reshape(-1, 12), reg_views=float(self.V),
                           opac_fac=self.opac_fac * frac, scale_fac=self.scale_fac * frac, out=self.grads)
        # regulariser part of the loss for the own Gaussians, added once per view like the reference (gs.py:150-152)
        loss_out += (self.reg[:2] * self.kreg).sum().float()
        self.t += 1
        ops.adam_step(self.ctx, P, self.grads, self.m, self.v, self.lr, 0.9, 0.999, 1e-8, self.t)
        return st
