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
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


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
