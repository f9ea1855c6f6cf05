"""3D Gaussian Splatting refinement -- drop-in for the reference's `starster.gs`
(starster/gs.py:14-166; docs/api.rst "3DGS refinement"): same function names, arguments, defaults,
return values and `scene.*` attributes, with every per-iteration operation running in
libst3r_hip.so (no gsplat, no torchmetrics, no autograd graph in the training loop).

Reference quirks that are reproduced on purpose: scales/opacities are rendered RAW
(gs.py:79-80) while the regularisers treat them as log / logit (gs.py:132,134); sh0 is allocated and
given an optimiser but never rendered (gs.py:25,29 vs :81); colours are initialised as 1 - colour in every
SH row (gs.py:29-31); both regularisers are added once per view (gs.py:150-152); `step` restarts at 0 on
every run_3dgs_optim call (gs.py:143) while the Adam step counters persist.
"""
__all__ = (
    "init_3dgs",
    "render_3dgs",
    "render_3dgs_original",
    "run_3dgs_optim",
    "train",
)

import numpy as np
import torch

from . import dist as _dist
from . import ops
from . import _lib as _lib_mod

GAUSSIAN_KEYS = ("means", "scales", "quats", "opacities", "sh0", "shN")
ADAM_BLOCKS = (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 1), ("shN", 12))  # block layout of [23N]


class FusedAdam:
    """Stands where the reference keeps one torch.optim.Adam per tensor (gs.py:37): exposes the slice of the
    fused optimiser state (exp_avg, exp_avg_sq, step) that belongs to one parameter."""

    def __init__(self, owner, key):
        self._owner, self.key = owner, key

    @property
    def param_groups(self):  # always the live parameter (growth replaces the tensors)
        o = self._owner
        return [dict(params=[o.scene.gaussians[self.key]], lr=o.lr, betas=o.betas, eps=o.eps)]

    @property
    def state(self):
        o = self._owner
        if self.key == "sh0":  # never receives a gradient in the reference (grad is None): no state
            return {}
        sl = o.block_slice(self.key)
        return {self.param_groups[0]["params"][0]: dict(step=o.step, exp_avg=o.m[sl], exp_avg_sq=o.v[sl])}

    def step(self):
        raise RuntimeError("the fused optimiser is stepped by run_3dgs_optim (st3r_adam_step), not per tensor")

    def zero_grad(self, set_to_none=True):
        pass


class _OptimState:
    def __init__(self, scene, lr):
        N = scene.gaussians["means"].shape[0]
        dev = scene.gaussians["means"].device
        self.scene, self.lr, self.betas, self.eps, self.step, self.N = scene, lr, (0.9, 0.999), 1e-8, 0, N
        self.m = torch.zeros(23 * N, device=dev); self.v = torch.zeros(23 * N, device=dev)
        self.grads = torch.empty(23 * N, device=dev)

    def grow(self, n_total):
        """Zero-extend the moments to n_total Gaussians (block layout: every block moves)."""
        N = self.N
        m = torch.zeros(23 * n_total, device=self.m.device); v = torch.zeros_like(m)
        off = 0
        for _, w in ADAM_BLOCKS:
            m[off * n_total:off * n_total + w * N].copy_(self.m[off * N:(off + w) * N])
            v[off * n_total:off * n_total + w * N].copy_(self.v[off * N:(off + w) * N])
            off += w
        self.m, self.v, self.N = m, v, n_total
        self.grads = torch.empty(23 * n_total, device=m.device)

    def block_slice(self, key):
        off = 0
        for k, w in ADAM_BLOCKS:
            if k == key:
                return slice(off * self.N, (off + w) * self.N)
            off += w
        raise KeyError(key)


class SSIM:
    """Callable stand-in for torchmetrics StructuralSimilarityIndexMeasure(data_range=1) (gs.py:39):
    `ssim(preds, target)` with (B,3,H,W) tensors returns the mean SSIM, computed by st3r_loss_l1_ssim."""

    def __init__(self, device):
        self.device = torch.device(device)

    def to(self, device):
        self.device = torch.device(device)
        return self

    def __call__(self, preds, target):
        ctx = ops.get_context(self.device)
        x = preds.permute(0, 2, 3, 1).contiguous().float(); y = target.permute(0, 2, 3, 1).contiguous().float()
        sums, _ = ops.loss_l1_ssim(ctx, x, y, 1.0, 0.0, want_grad=False)
        H, W = x.shape[1], x.shape[2]
        return (sums[:, 1] / ((H - 10) * (W - 10) * 3)).mean().float()


class MCMCStrategy:
    """Stands where the reference constructs gsplat.MCMCStrategy() (gs.py:43-45) with its default
    hyper-parameters; the three refinement operations run in libst3r_hip.so (csrc/mcmc.hip):

      step_post_backward(step, lr): inside the refine window (500 < step < 25000, every 100 steps) dead
      Gaussians are relocated onto alive ones and the set grows by 5 % up to cap_max; position noise is
      injected on every call.  Opacities/scales are read as logits/logs here while the renderer uses the
      same tensors raw -- the reference's behaviour.

    Random draws come from a counter-based generator keyed by state["seed"] and a call counter, not from
    torch's global generator: view-sharded replicas take identical decisions with no communication."""
    cap_max = 1_000_000; noise_lr = 5e5; refine_start_iter = 500; refine_stop_iter = 25_000
    refine_every = 100; min_opacity = 0.005

    def check_sanity(self, params, optimizers):
        for k in ("means", "scales", "quats", "opacities"):
            assert k in params and k in optimizers, f"{k} is required"

    def initialize_state(self, seed=0):
        return {"binoms": None, "seed": int(seed), "calls": 0, "n_relocated": 0, "n_added": 0}

    def step_pre_backward(self, params, optimizers, state, step, info):
        return None

    def is_refine_step(self, step):
        return self.refine_start_iter < step < self.refine_stop_iter and step % self.refine_every == 0

    def refine(self, params, optimizers, state, step, call):
        """relocate dead Gaussians + grow by 5 % (needs every Gaussian: the full, replicated tensors)."""
        owner = optimizers["means"]._owner
        ctx = ops.get_context(params["means"].device)
        seed = state.get("seed", 0)
        P = {k: params[k].data for k in GAUSSIAN_KEYS}
        state["n_relocated"] = ops.mcmc_relocate(ctx, P, owner.m, owner.v, self.min_opacity, seed, call)
        N = P["means"].shape[0]
        n_new = max(0, min(self.cap_max, int(1.05 * N)) - N)
        if n_new > 0:
            grown = {}
            for k in GAUSSIAN_KEYS:
                t = torch.empty((N + n_new,) + tuple(P[k].shape[1:]), dtype=P[k].dtype, device=P[k].device)
                t[:N].copy_(P[k])
                grown[k] = t
            ops.mcmc_add(ctx, grown, N, n_new, self.min_opacity, seed, call)
            for k in GAUSSIAN_KEYS:  # new leaf tensors, as gsplat re-creates the nn.Parameters
                params[k] = torch.nn.Parameter(grown[k], requires_grad=True)
            owner.grow(N + n_new)
        state["n_added"] = n_new

    def step_post_backward(self, params, optimizers, state, step, info, lr):
        ctx = ops.get_context(params["means"].device)
        seed, call = state.get("seed", 0), state.get("calls", 0)
        state["calls"] = call + 1
        with torch.no_grad():
            if self.is_refine_step(step):
                self.refine(params, optimizers, state, step, call)
            P = {k: params[k].data for k in ("means", "quats", "scales", "opacities")}
            ops.mcmc_noise(ctx, P, lr * self.noise_lr, seed, call)


def init_3dgs(scene, init_scale=3e-3, lr=1e-3):
    """Initialize 3DGS splats and optims from Mast3r dense points (reference gs.py:14-45)."""
    pts = scene.dense_pts_flat
    colors = scene.dense_cols_flat
    n = pts.shape[0]
    g = {
        "means": pts.clone().float(),
        "scales": torch.full_like(pts, init_scale, dtype=torch.float32),
        "quats": torch.zeros(n, 4),
        "opacities": torch.ones(n),
        "sh0": torch.zeros(n, 1, 3),
        "shN": torch.zeros(n, 24, 3),
    }
    g["quats"][:, 0] = 1.0
    inv = (1 - colors).float()
    g["sh0"][:, 0] = inv
    g["shN"][:] = inv[:, None, :]
    scene.gaussians = {k: torch.nn.Parameter(v.to(scene.device).contiguous(), requires_grad=True) for k, v in g.items()}
    scene._gs_optim = _OptimState(scene, lr)
    scene.optimizers = {k: FusedAdam(scene._gs_optim, k) for k in scene.gaussians}
    scene.ssim = SSIM(scene.device)
    scene.strategy = MCMCStrategy()
    scene.strategy.check_sanity(scene.gaussians, scene.optimizers)
    scene.strategy_state = scene.strategy.initialize_state()
    scene._gt_dev = None


class _Rasterize(torch.autograd.Function):
    """Differentiable wrapper so user code can still back-propagate through render_3dgs."""

    @staticmethod
    def forward(fctx, means, quats, scales, opacities, shN, w2c, Ks, width, height, ctx):
        rgb, alpha, info = ops.rasterization(ctx, means, quats, scales, opacities, shN, w2c, Ks, width, height)
        fctx.save_for_backward(means, quats, scales, opacities, shN, w2c, Ks, alpha)
        fctx.info, fctx.ctx, fctx.wh = info, ctx, (width, height)
        return rgb, alpha

    @staticmethod
    def backward(fctx, v_rgb, v_alpha):
        means, quats, scales, opacities, shN, w2c, Ks, alpha = fctx.saved_tensors
        info, ctx, (W, H) = fctx.info, fctx.ctx, fctx.wh
        Cn = w2c.shape[0]
        # the backward consumes the contribution masks the forward left in the ctx: re-run the blend forward so
        # that they belong to THIS rasterization even if other renders happened in between
        ops.blend_fwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], Cn, W, H)
        v_splats = ops.blend_bwd(ctx, info["_splats"], info["isect_offsets"], info["_flatten_ids_dense"], alpha,
                                 info["_last_ids"], v_rgb.contiguous().float(),
                                 None if v_alpha is None else v_alpha.contiguous().float(), info["_cum_tiles"], Cn, W, H)
        grads = ops.project_sh_bwd(ctx, means, quats, scales, opacities, shN, w2c, Ks, info["_campos"], W, H,
                                   info["_splats"], v_splats)
        G = ops.split_grads(grads, means.shape[0])
        v_sh = torch.zeros_like(shN)
        v_sh[:, :4] = G["sh"]
        return G["means"], G["quats"], G["scales"], G["opacities"], v_sh, None, None, None, None, None


def render_3dgs(scene, w2c: torch.Tensor, intrinsics: torch.Tensor, width: int, height: int):
    """Render the splats from a set of camera views (reference gs.py:47-88).

    Returns the tuple the reference gets from gsplat.rasterization: (render_img (N,H,W,3),
    render_alpha (N,H,W,1), info)."""
    g = scene.gaussians
    ctx = ops.get_context(scene.device)
    w2c = w2c.to(scene.device, torch.float32).contiguous(); Ks = intrinsics.to(scene.device, torch.float32).contiguous()
    rgb, alpha = _Rasterize.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, width,
                                  height, ctx)
    # the info dict of the most recent rasterization (gsplat's `meta`)
    info = {k: v for k, v in ops.last_info().items() if not k.startswith("_")}
    return rgb, alpha, info


def render_3dgs_original(scene, width: int, height: int):
    """Render from camera views of original scene (reference gs.py:90-95)."""
    return scene.render_3dgs(scene.w2c, scene.intrinsics, width, height)


def _gt_on_device(scene, views):
    """GT images are uploaded once and cached (the reference re-uploads every view every iteration, gs.py:151)."""
    key = (tuple(views), len(scene.imgs))
    if getattr(scene, "_gt_dev", None) is None or scene._gt_dev[0] != key:
        imgs = [torch.as_tensor(np.asarray(scene.imgs[i]), dtype=torch.float32) for i in views]
        scene._gt_dev = (key, torch.stack(imgs).to(scene.device).contiguous())
        scene._gt_mom = None
    return scene._gt_dev[1]


def _gt_moments(scene, ctx, gt):
    """SSIM's windowed moments of the ground truth, conv(gt) and conv(gt^2): the reference recomputes them every iteration
    (torchmetrics, gs.py:129) although the images never change; here once per uploaded image set."""
    if getattr(scene, "_gt_mom", None) is None or scene._gt_mom[0] is not gt:
        scene._gt_mom = (gt, ops.gt_moments(ctx, gt))
    return scene._gt_mom[1]


def run_3dgs_optim(
        scene,
        iters: int,
        enable_pruning: bool = False,
        loss_ssim_fac=0.2,
        loss_opacity_fac=0.01,
        loss_scale_fac=0.01,
        verbose: bool = False,
    ) -> list:
    """Run 3DGS optimization and pruning (optional) for a number of iterations (reference gs.py:97-166).

    Returns the list of per-iteration losses (floats).  Under torch.distributed (one process per GPU, RCCL)
    the views are sharded over the ranks, the [23N] gradient buffer is sum-all-reduced every iteration and
    every rank applies the identical fused Adam update; the returned losses are the sums over all views.
    """
    height, width = scene.imgs[0].shape[:2]
    ctx = ops.get_context(scene.device)
    st = scene._gs_optim
    g = scene.gaussians
    rank, world = _dist.rank_world()
    if _sharded_layout(scene, world, enable_pruning):
        return _run_3dgs_optim_sharded(scene, iters, loss_ssim_fac, loss_opacity_fac, loss_scale_fac, verbose,
                                       enable_pruning)
    views = _dist.shard_views(len(scene.imgs), rank, world)
    w2c_all = scene.w2c.to(scene.device, torch.float32)
    w2c = w2c_all[views].contiguous()
    Ks = scene.intrinsics.to(scene.device, torch.float32)[views].contiguous()
    campos = ops.camera_positions(w2c)
    gt = _gt_on_device(scene, views)
    losses = torch.zeros(max(iters, 1), device=scene.device)
    fused = world == 1 or getattr(ctx, "native_comm", False)
    restore_exchange = None
    if enable_pruning and fused and world > 1 and ops.get_exchange(ctx) in ("rs_ag", "direct"):
        # reduce-scatter exchange (through RCCL or through the peers' exported buffers): a rank maintains the Adam moments of its piece of the buffer only; growing the set
        # moves the piece boundaries, so a refinement run uses the plain all-reduce (replicated moments) for its duration:
        # the pieces are all-gathered first so that every rank holds the same, complete moments
        ops.allgather_pieces(ctx, st.m); ops.allgather_pieces(ctx, st.v)
        restore_exchange = ops.set_exchange(ctx, "allreduce")
    it_range = range(iters)
    if verbose:
        from tqdm import trange
        it_range = trange(iters)
    # The kernels touch SH rows 0..3 only (sh_degree = 1, gs.py:81,86).  Read in place those 48 bytes sit at a 288-byte
    # stride -- three streaming kernels (projection, its backward, Adam) then move ~2.7x the SH bytes they use --, so the loop
    # trains a COMPACT copy [N, 4, 3] (sh_stride = 12) and writes it back into rows 0..3 of `shN` when it ends, and before
    # every strategy hook that may look at the parameters (measured at 1 M Gaussians: projection 0.183 -> 0.157 ms, its
    # backward 0.223 -> 0.203, Adam 0.148 -> 0.105; tools/experiments/ab_compact_sh.sh).
    sh_c = [g["shN"].data[:, :4].contiguous()]

    def sh_write_back():
        g["shN"].data[:, :4].copy_(sh_c[0])

    def one_iteration(step):
        if sh_c[0].shape[0] != g["shN"].shape[0]:   # the set grew (a refinement step replaced the tensors)
            sh_c[0] = g["shN"].data[:, :4].contiguous()
        P = {k: g[k].data for k in ("means", "quats", "scales", "opacities")}  # growth replaces the tensors
        P["shN"] = sh_c[0]
        if fused:   # the whole iteration is one C call (gradient all-reduce inside, over the ctx's communicator)
            # no host round trip in steady state (single rank; with a communicator the library sizes every step exactly)
            ops.train_step(ctx, P, w2c, Ks, campos, gt, width, height, loss_ssim_fac, loss_opacity_fac,
                           loss_scale_fac, st.grads, st.m, st.v, st.lr, st.betas[0], st.betas[1], st.eps, st.step,
                           losses[step:step + 1], want_stats=False)
        else:       # gradient all-reduce through the host framework's process group (torch.distributed -> RCCL); every
                    # step is sized exactly (want_stats): an overflow seen by one rank only would leave the others waiting
            ops.train_fwd_bwd(ctx, P, w2c, Ks, campos, gt, width, height, loss_ssim_fac, loss_opacity_fac,
                              loss_scale_fac, st.grads, losses[step:step + 1], want_stats=True)
            _dist.all_reduce_sum(st.grads)
            ops.adam_step(ctx, P, st.grads, st.m, st.v, st.lr, st.betas[0], st.betas[1], st.eps, st.step)

    def capacity_error(e):
        return getattr(e, "code", 0) == -3 and world == 1

    # A peer failure (ST3R_ERR_PEER: the step before failed on some rank, nobody applied it) ABORTS the run on every rank
    # alike -- all ranks get the code from the same call --; the exchange form is restored whatever ends the loop.
    # our own strategy reads the parameters in step_post_backward only, and the SH rows on refinement steps only
    # (relocation / growth copy whole rows); any OTHER strategy object sees up-to-date rows at every hook call
    own_strategy = type(scene.strategy) is MCMCStrategy
    ops.set_gt_moments(ctx, gt, _gt_moments(scene, ctx, gt))
    try:
        step = 0
        for _ in it_range:
            if enable_pruning:
                if not own_strategy:
                    sh_write_back()
                scene.strategy.step_pre_backward(g, scene.optimizers, scene.strategy_state, step, None)
                if not own_strategy:
                    sh_c[0] = g["shN"].data[:, :4].contiguous()
            st.step += 1
            try:
                one_iteration(step)
            except _lib_mod.St3rError as e:
                # ST3R_ERR_CAPACITY is reported by the call AFTER the asynchronous step that outgrew its buffers (> 25 % more
                # tile intersections than the step before it).  That step's records past the capacity were dropped and its
                # Adam update was skipped on the device (k_adam's guard), so no parameter update has to be undone: the lost
                # iteration is repeated -- the context is back on the exactly sized path -- and then this one runs.  With
                # enable_pruning the strategy hooks of the lost iteration have already run on the un-updated parameters
                # (its position noise, and on a refinement step its relocation / growth): they are NOT run again, i.e. the
                # noise of that one iteration is drawn before its update instead of after it.  A deviation of one step's
                # noise (tests/test_gpu_api.py::test_run_3dgs_optim_repeats_...); overflows need > 25 % more tile
                # intersections than the step before.
                if not capacity_error(e):
                    raise
                if step > 0:
                    st.step -= 1
                    one_iteration(step - 1)
                    st.step += 1
                one_iteration(step)
            if enable_pruning:
                touch = bool(scene.strategy.is_refine_step(step)) if own_strategy else True
                if touch:
                    sh_write_back()
                scene.strategy.step_post_backward(g, scene.optimizers, scene.strategy_state, step, None, 1e-3)
                if touch:
                    sh_c[0] = g["shN"].data[:, :4].contiguous()
            step += 1
        if world == 1 and iters > 0:
            try:   # the last step's count is still in flight: settle it now so that an overflow cannot go unnoticed
                ops.settle(ctx)
            except _lib_mod.St3rError as e:
                if not capacity_error(e):
                    raise
                one_iteration(iters - 1)   # its update was skipped on the device: repeat it
    finally:
        ops.set_gt_moments(ctx, None, None)
        if sh_c[0].shape[0] == g["shN"].shape[0]:
            sh_write_back()
        if restore_exchange is not None:   # (the moments stay complete on every rank: rs_ag goes on using its own piece)
            ops.set_exchange(ctx, restore_exchange)
    _dist.all_reduce_sum(losses)
    return losses[:iters].cpu().tolist()   # one device->host copy for the whole call (reference: .item() per step)


def _sharded_layout(scene, world, enable_pruning):
    """Gaussians AND views sharded (layout 2) is OPT-IN: ST3R_MULTI_GPU=gaussian-sharded (also with
    a single rank: tests).  The default under torch.distributed is the north_star partition -- views sharded,
    Gaussians replicated, one gradient all-reduce per iteration.  With enable_pruning the sharded loop re-assembles
    the full tensors for the refinement steps (every 100 iterations) and shards the grown set again."""
    import os
    if os.environ.get("ST3R_MULTI_GPU", "replicated") != "gaussian-sharded":
        return False
    return len(scene.imgs) % world == 0 and scene.gaussians["means"].shape[0] >= world


def _gather_rows(full, local, counts, width):
    """full [N * width] (rows of `width` floats, rank-ordered shards) <- every rank's local rows."""
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        if all(c == counts[0] for c in counts):
            tdist.all_gather_into_tensor(full.reshape(-1), local.reshape(-1).clone())   # local may be a view of full
        else:
            parts = _dist.all_gather_varlen(local.reshape(-1).clone())
            torch.cat([p.to(full.device) for p in parts], out=full.reshape(-1))
    else:
        full.reshape(-1).copy_(local.reshape(-1))


def _run_3dgs_optim_sharded(scene, iters, ssim_fac, opac_fac, scale_fac, verbose, enable_pruning=False):
    """run_3dgs_optim on the Gaussian-sharded layout: this rank trains rows [lo, hi) of the (replicated) parameter
    tensors in place, exchanging splat records with the other ranks; at the end the shards are all-gathered so that
    scene.gaussians and the optimiser state are complete on every rank again.

    enable_pruning: the position noise of every iteration is drawn per shard (keyed by the global row, so it equals
    the replicated draw); on a refinement step the shards are gathered, relocate / grow run on the full tensors --
    identically on every rank, like in the replicated layout -- and the grown set is sharded again (balanced, sizes
    differing by at most one).  Between refinement steps nothing is replicated."""
    height, width = scene.imgs[0].shape[:2]
    ctx = ops.get_context(scene.device)
    st = scene._gs_optim
    rank, world = _dist.rank_world()
    views = _dist.shard_views_contiguous(len(scene.imgs), rank, world)
    w2c_all = scene.w2c.to(scene.device, torch.float32).contiguous()
    Ks_all = scene.intrinsics.to(scene.device, torch.float32).contiguous()
    gt = _gt_on_device(scene, views)
    keys = ("means", "quats", "scales", "opacities", "shN")

    def shard():
        """trainer over this rank's rows of the current full tensors (views into them: updated in place)"""
        g = scene.gaussians
        N = g["means"].shape[0]
        counts = _dist.shard_counts(N, world)
        lo, hi = _dist.shard_gaussians(N, rank, world)
        n = hi - lo
        P = {k: g[k].data[lo:hi] for k in keys}
        tr = _dist.ShardedTrainer(ctx, P, N, w2c_all, Ks_all, gt, width, height, rank, world, lr=st.lr,
                                  ssim_fac=ssim_fac, opac_fac=opac_fac, scale_fac=scale_fac, counts=counts)
        off = 0
        for _, w in ADAM_BLOCKS:                                    # this shard's rows of the [23N] moment blocks
            tr.m[off * n:(off + w) * n].copy_(st.m[off * N + w * lo:off * N + w * hi])
            tr.v[off * n:(off + w) * n].copy_(st.v[off * N + w * lo:off * N + w * hi])
            off += w
        tr.t = st.step
        return tr, P, N, n, lo, counts

    def unshard(tr, P, N, n, counts):
        """every rank's rows back into the full tensors and moment blocks"""
        g = scene.gaussians
        st.step = tr.t
        for k in keys:
            _gather_rows(g[k].data, P[k], counts, g[k].data[0].numel())
        off = 0
        for _, w in ADAM_BLOCKS:
            _gather_rows(st.m[off * N:(off + w) * N], tr.m[off * n:(off + w) * n], counts, w)
            _gather_rows(st.v[off * N:(off + w) * N], tr.v[off * n:(off + w) * n], counts, w)
            off += w

    tr, P, N, n, lo, counts = shard()
    losses = torch.zeros(max(iters, 1), device=scene.device)
    it_range = range(iters)
    if verbose:
        from tqdm import trange
        it_range = trange(iters)
    strategy, state = scene.strategy, scene.strategy_state
    ops.set_gt_moments(ctx, gt, _gt_moments(scene, ctx, gt))
    try:
        for step in it_range:
            tr.step(losses[step:step + 1])
            if enable_pruning:   # MCMCStrategy.step_post_backward, shard-wise
                seed, call = state.get("seed", 0), state.get("calls", 0)
                state["calls"] = call + 1
                with torch.no_grad():
                    if strategy.is_refine_step(step):
                        unshard(tr, P, N, n, counts)
                        strategy.refine(scene.gaussians, scene.optimizers, state, step, call)
                        tr, P, N, n, lo, counts = shard()
                    ops.mcmc_noise(ctx, {k: P[k] for k in ("means", "quats", "scales", "opacities")},
                                   1e-3 * strategy.noise_lr, seed, call, row_offset=lo)
    finally:
        ops.set_gt_moments(ctx, None, None)
    unshard(tr, P, N, n, counts)
    _dist.all_reduce_sum(losses)
    return losses[:iters].cpu().tolist()


train = run_3dgs_optim  # alias for the wording of BASELINE.json's north_star ("gs.train()")
