#!/usr/bin/env python
"""Headline benchmark: 3DGS train iters/sec @ 1M Gaussians, 8 x 1080p views (BASELINE.json).

One "step" = one iteration of the reference loop starster/gs.py:143-161: render every view,
L1+SSIM loss (+ regularisers), backward, Adam.  Workload = SYNTH-1M of SURVEY.md 8(d)
(BASELINE.json configs[2]); synthetic data, random-init parameters.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the 8 views are sharded 8/N per rank (strong scaling of the named config), Gaussians
and Adam state are replicated, one RCCL sum-all-reduce of the [23*N] gradient buffer per step.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     -- dominant kernel: algorithmic bytes per launch / mean launch time, the launch
                  time measured with HIP events on the launch stream inside the timed region
  cpu_baseline -- the C oracle (oracle/gs_oracle.c, a "port") timed on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X datasheet HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # >= 200 iterations, median of 5 windows
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-drift", action="store_true",
                    help="skip the untimed continuation to 200 steps that shows how the step time drifts as training "
                         "changes the scene (windows.drift; one GPU, only when --steps < 200)")
    ap.add_argument("--no-scaling-model", action="store_true",
                    help="skip the single-GPU model of the 1 -> 8 GPU curve (scaling_model: this rank's step at 8 / 4 / 2 / 1 "
                         "views + the one-rank exchange), ~1 s")
    ap.add_argument("--no-config1", action="store_true",
                    help="skip the labelled secondary measurement at BASELINE configs[1]'s size (profiling runs: its 75 steps of "
                         "a smaller workload would enter the per-kernel means)")
    ap.add_argument("--train-only", action="store_true",
                    help="skip the alignment / matching / condensation benches behind the headline (profiling runs)")
    ap.add_argument("--multi-gpu", choices=("replicated", "gaussian-sharded"), default="replicated",
                    help="how N > 1 GPUs are used (DESIGN.md section 5): replicated = the north_star partition (default)")
    ap.add_argument("--cpu-sample-div", type=int, default=2,
                    help="CPU baseline sample: 1 view at (W/div)x(H/div), N/div^2 gaussians, same density")
    return ap.parse_args()


def algorithmic_bytes_reference(N, V, I, P, Ct, keybits):
    """SURVEY.md 8(d) as written: algorithmic HBM bytes per iteration of the REFERENCE algorithm (gsplat's 3-sigma
    tile squares, I = its intersection count, one 64-bit-key LSD sort).  Reported as `vs_reference_algorithm` only:
    the fused path never touches the I - I_kept culled records, and bytes not moved are not achieved bandwidth."""
    passes = math.ceil(keybits / 8)
    return {
        "project": 92 * N + 44 * V, "scan": 8 * V, "emit": 8 * V + 12 * I, "sort": 24 * passes * I,
        "offsets": 8 * I + 4 * Ct, "blend_fwd": 40 * I + 20 * P, "loss": 36 * P, "blend_bwd": 112 * I + 20 * P,
        "project_bwd": 80 * V + 92 * N, "adam": 644 * N,
    }


def algorithmic_bytes(N, C, V, I_kept, P, Ct, key1_bits, key1_bytes, key2_bits):
    """Algorithmic HBM bytes per iteration of what the fused path EXECUTES: the per-unit figures of SURVEY.md 8(d)
    (92 B / Gaussian and 44 B / visible pair for the projection, 40 B / record and 20 B / pixel for the blend
    forward, 112 B / record and 20 B / pixel for the blend backward, 36 B / pixel for the loss, ...) times the units a
    launch processes -- records = I_kept, the (record, tile) pairs that survive the exact alpha >= 1/255 culling --
    and, for the two sorts, the passes the two-level sort runs: level 1 sorts all N*C pair slots with
    (key1_bytes + 4)-byte items in ceil(key1_bits / 8) passes, level 2 the I_kept records with 8-byte items in
    ceil(key2_bits / 8) passes; a pass reads and writes every item once, the histogram launch reads the keys once."""
    n_pairs = N * C
    p1, p2 = math.ceil(key1_bits / 8), math.ceil(key2_bits / 8)
    return {
        "project": 92 * N + 44 * V,
        "scan": 8 * V,
        "emit": 8 * V + 12 * I_kept,
        "sort_depth": (2 * p1 * (key1_bytes + 4) + key1_bytes) * n_pairs,
        "sort": (2 * p2 * 8 + 4) * I_kept,
        "offsets": 8 * I_kept + 4 * Ct,
        "blend_fwd": 40 * I_kept + 20 * P,
        "loss": 36 * P,
        "blend_bwd": 112 * I_kept + 20 * P,
        "project_bwd": 80 * V + 92 * N,
        "adam": 644 * N,
    }


def make_gt_images(ctx, ops, g_np, w2c, Ks, W, H, device):
    """GT = render of the jittered scene, clipped to [0,1]; produced on device."""
    from st3r_synth import synth
    gt_g = synth.perturb_for_gt(g_np)
    P = {k: torch.tensor(v, device=device) for k, v in gt_g.items()}
    campos = ops.camera_positions(w2c)
    rgb, _, _ = ops.render(ctx, P, w2c, Ks, campos, W, H)
    return rgb.clamp_(0, 1).contiguous()


def cpu_baseline(args):
    """Time the C oracle (single thread) on a bounded sample and extrapolate to the metric's unit.
    Sample: three views at (W/d)x(H/d) with N/d^2 Gaussians (~12 s): the scene extent and the cameras are unchanged, the
    focal scales with the image, so the per-pixel list length drops by d^2 too; the extrapolation is therefore reported
    as measured-sample only."""
    from oracle import build as ob
    ob.build()
    from oracle import gs_oracle as go
    from st3r_synth import synth
    d = args.cpu_sample_div
    W, H, N = args.width // d, args.height // d, args.gaussians // (d * d)
    # three views of the sample scene, one after the other (~12 s of CPU work: the contract asks for 10 - 30 s)
    SV = 3
    g, w2c_all, Ks_all = synth.make_scene(N, SV, W, H)
    gt_g = synth.perturb_for_gt(g)
    t = 0.0
    for sv in range(SV):
        w2c, Ks = w2c_all[sv:sv + 1], Ks_all[sv:sv + 1]
        gt_img, _, _ = go.rasterization(gt_g["means"], gt_g["quats"], gt_g["scales"], gt_g["opacities"], gt_g["shN"], w2c,
                                        Ks, W, H)
        gt_img = np.clip(gt_img, 0, 1)
        t0 = time.perf_counter()
        rgb, alpha, meta = go.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H)
        _, _, v_rgb = go.l1_ssim(rgb[0], gt_img[0], 0.8, 0.2)
        grads = go.rasterization_backward(g["means"], g["quats"], g["scales"], g["opacities"], g["shN"], w2c, Ks, W, H,
                                          meta, alpha, v_rgb[None])
        p = g["means"].reshape(-1).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
        for _ in range(23 // 3 + 1):  # Adam over ~23 scalars per gaussian
            go.adam(p, grads["means"].astype(np.float32).reshape(-1), m, v, 1e-3, 0.9, 0.999, 1e-8, 1)
        t += time.perf_counter() - t0
    total_s = t
    t = t / SV   # seconds per sample view
    # one full iteration = views x d^2 (pixels) samples of this size; d^2 more gaussians per pixel list too
    scale = args.views * d * d
    return {
        "value": 1.0 / (t * scale), "unit": "iters/sec", "cores": 1, "kind": "port",
        # (kept under 128 characters: the driver's record truncates longer strings)
        "sample": f"C oracle, 1 thread: {SV} views {W}x{H}, {N} gaussians, fwd+loss+bwd+Adam {t:.2f}s/view; "
                  f"value=1/(t*{scale}), area-scaled",
        "sample_note": f"scaled by views * {d * d} (pixel area) only: optimistic for the CPU, the full scene also has "
                       f"{d * d}x more gaussians per pixel list",
        "sample_seconds": total_s,
    }


def align_bench(device, with_cpu=True, views=8, cpu_iters=(100, 40)):
    """Secondary metric "align sec": wall time of the reference's 500+200-iteration global alignment
    (starster/reconstruct.py:427,440) on a synthetic condensed problem -- HIP path vs the torch-CPU oracle
    (a port of the reference loop, validated against reference-generated goldens) on the host cores."""
    from starst3r_amd import align
    from st3r_synth import synth_align
    flat = synth_align.flatten(synth_align.make_problem(n_views=views, n_corr=2000 // (views - 1) + 1, seed=1))
    align.run(flat, niter1=3, niter2=3, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _ = align.run(flat, device=device)
    torch.cuda.synchronize()
    hip_s = time.perf_counter() - t0
    L = res["losses"].cpu().numpy()
    out = {"views": views, "anchors": int(flat["anchor_idx"].size), "correspondence_rows": int(flat["corr_a1"].size),
           "iterations": "500+200", "hip_seconds": hip_s, "hip_ms_per_iter": hip_s / 700 * 1e3,
           "loss_stage1": [float(L[0]), float(L[499])], "loss_stage2": [float(L[500]), float(L[-1])],
           "bound": "latency of dependent steps, not launches (measured, tools/align_profile.py): ~12 us residual kernel + ~14 us one-workgroup update kernel (sequential walks over the MST) per iteration; working set in L2 -- not roofline bound"}
    if with_cpu:
        from oracle import align_oracle
        # The loop is ~6400 tiny ATen ops per iteration: more intra-op threads only add
        # synchronisation cost (256 threads on the GPU box's host: >20 minutes).  8 threads is the setting the
        # survey measured (7.3-7.9 s on 8 Xeon cores); a 100+40-iteration sample keeps the run bounded and is
        # scaled to the full 500+200 schedule (cost per iteration is constant).
        cores = min(8, os.cpu_count() or 1)
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(cores)
        n1, n2 = cpu_iters
        t0 = time.perf_counter()
        align_oracle.run(flat, niter1=n1, niter2=n2)
        cpu_sample = time.perf_counter() - t0
        torch.set_num_threads(prev_threads)
        cpu_s = cpu_sample * (500.0 / n1)
        out.update(cpu_port_seconds=cpu_s, cpu_cores=cores, cpu_kind="port", cpu_sample_seconds=cpu_sample,
                   cpu_note="oracle/align_oracle.py (torch CPU autograd restatement of reconstruct.py:116-457), "
                            f"{cores} intra-op threads, {n1}+{n2} iterations timed and scaled x{500 // n1} to 500+200",
                   speedup=cpu_s / hip_s)
    return out


def condense_bench(device, with_cpu=True, views=8, W=512, H=384):
    """canonical pointmaps + focals + anchors for `views` images of W x H from the C(C-1)/2 pair
    predictions (the step between matching and alignment), device kernels vs the numpy restatement."""
    from starst3r_amd import condense
    from st3r_synth import synth_pairs
    P = synth_pairs.make_pair_predictions(views, W, H, seed=0, n_corr=2000)
    dev_pairs = {}
    for k, ((p1, p2), (score, corr)) in P["pairs"].items():   # resident on the device like freshly inferred pairs
        dev_pairs[k] = ((tuple(torch.tensor(a, device=device) for a in p1), tuple(torch.tensor(a, device=device) for a in p2)),
                        (score, tuple(torch.tensor(a, device=device) for a in corr)))
    w = condense.prepare_canonical_data(P["imgs"], dev_pairs, 8, device=device)     # warm-up of both parts
    condense.condense_data(P["imgs"], dev_pairs, w[2], w[4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, scores, cviews, _, preds = condense.prepare_canonical_data(P["imgs"], dev_pairs, 8, device=device)
    torch.cuda.synchronize()
    t_canon = time.perf_counter() - t0
    t0 = time.perf_counter()
    condense.compute_min_spanning_tree(scores)
    condense.condense_data(P["imgs"], dev_pairs, cviews, preds)
    torch.cuda.synchronize()
    t_lists = time.perf_counter() - t0
    f = [float(cviews[i][2][0]) for i in P["imgs"]]
    out = {"views": views, "image": f"{W}x{H}", "pairs": len(dev_pairs), "prepare_canonical_data_ms": t_canon * 1e3,
           "mst_and_condense_data_ms": t_lists * 1e3, "focal_error_max": max(abs(x - P["focal_true"]) for x in f) / P["focal_true"],
           "bound": "HBM / latency, small (not roofline bound)"}
    if with_cpu:
        from oracle import condense_oracle as co
        t0 = time.perf_counter()
        for img in P["imgs"][:2]:
            pt, cf = [], []
            for (a, b), ((p1, p2), _c) in P["pairs"].items():
                if a == img:
                    pt.append(p1[0]); cf.append(p1[1])
                elif b == img:
                    pt.append(p2[0]); cf.append(p2[1])
            canon, canon2, _ = co.canonical_view(np.stack(pt), np.stack(cf), 8)
            co.estimate_focal_knowing_depth(canon, (W / 2, H / 2))
        cpu = (time.perf_counter() - t0) * views / 2
        out.update(cpu_port_ms=cpu * 1e3, cpu_kind="port", cpu_cores=1,
                   cpu_note="oracle/condense_oracle.py (numpy): canonical_view + focal of 2 images timed, scaled to all",
                   speedup=cpu / t_canon)
    return out


def csrc_fingerprint():
    """sha1 over the kernel sources: a PMC measurement belongs to the kernels it was taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "starst3r_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(stage, N, views, W, H, world):
    """HBM bytes per launch of the dominant stage from the PMC passes committed as profiles/pmc_traffic.json
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes; written by
    tools/pmc_summary.py together with the workload and the fingerprint of the kernel sources it was measured on).
    Counters cannot be read from inside the benchmark process, so the value is only reported when the file matches
    this workload AND these kernel sources; otherwise null (stale)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_traffic.json"
    rec = json.load(open(path))
    wl = rec.get("workload", {})
    if [wl.get(k) for k in ("gaussians", "views", "width", "height", "n_gpus")] != [N, views, W, H, world]:
        return None, "profiles/pmc_traffic.json was measured on another workload"
    if rec.get("csrc_fingerprint") != csrc_fingerprint():
        return None, f"stale: measured at commit {rec.get('commit')} on other kernel sources"
    v = rec.get("traffic_bytes_per_launch", {}).get(stage)
    return v, f"profiles/pmc_traffic.json @ {rec.get('commit')}: rocprofv3 --pmc, 2 x FETCH_SIZE + WRITE_SIZE per launch"


CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md)
# the clock the blend kernels were MEASURED to hold (GRBM_GUI_ACTIVE per XCD / launch duration, profiles/r5_clock_pmc.md):
# power management under their instruction mix, not a property of the schedule
MEASURED_CLOCK_GHZ = {"blend_fwd": 2.13, "blend_bwd": 2.10}


def valu_issue(per_stage_ms, N, views, W, H, world):
    """The roofline the blend kernels actually hit: VALU issue.  SQ_INSTS_VALU per launch (PMC pass of
    tools/profile_round4.sh, kept in profiles/pmc_traffic.json next to the traffic) x 2 cycles -- the guide's issue rate of
    a wave64 VALU instruction -- over 1024 SIMDs x clock x the kernel's time, and the same instruction count priced with
    the measured per-opcode issue costs of tools/probe/valu_cost.hip instead of a flat 2 cycles (adds / multiplies /
    fmac 1.0-1.1 ns, 3-source fma 1.47, compares / selects / min / max / DPP 1.7-1.8, exp / rcp 3.4 ns per instruction
    and SIMD: DESIGN.md section 4 derives 1.17 ns per instruction for the backward's mix, 1.30 ns for the forward's)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    wl = rec.get("workload", {})
    if [wl.get(k) for k in ("gaussians", "views", "width", "height", "n_gpus")] != [N, views, W, H, world]:
        return None
    insts = rec.get("valu_insts_per_launch")
    if not insts or rec.get("csrc_fingerprint") != csrc_fingerprint():
        return None
    share = rec.get("kernel_share_of_stage_time", {})
    # opcode-mix-weighted: only where DESIGN.md section 4 derives the kernel's budget -- the backward: 1.17 ns per VALU
    # instruction of its mix (phase 1: 23 VALU + exp + rcp per trip; phase 2: 34 plain + 20 DPP + 6 per four records)
    mix_ns = {"blend_bwd": 1.17}
    out = {"source": f"profiles/pmc_traffic.json (commit {rec.get('commit')}): SQ_INSTS_VALU per launch of k_blend_fwd_cells / "
                     "k_blend_bwd; kernel time = this run's stage time (HIP events) x the kernel's share of the stage in the "
                     "same profile; 1024 SIMDs at 2.4 GHz",
           "opcode_mix_source": "tools/probe/valu_cost.hip issue costs (add / mul / fmac 1.0-1.1 ns, 3-source fma 1.47, compares / "
                                "selects / min / max / DPP 1.7-1.8, exp / rcp 3.4 ns per instruction and SIMD) x the "
                                "instruction budget of DESIGN.md section 4"}
    for st in ("blend_fwd", "blend_bwd"):
        if st in insts and per_stage_ms.get(st):
            t = per_stage_ms[st] * 1e-3 * share.get(st, 1.0)
            per_simd = insts[st] / 1024.0
            out[st] = {"valu_insts_per_launch": insts[st], "kernel_ms": t * 1e3,
                       "frac_at_2_cycles": per_simd * 2.0 / (CLOCK_GHZ * 1e9) / t,
                       # the same at the clock the kernel actually ran at when it was profiled (not measured in THIS run)
                       "frac_at_2_cycles_at_profiled_clock": per_simd * 2.0 / (MEASURED_CLOCK_GHZ[st] * 1e9) / t,
                       "profiled_clock_ghz": MEASURED_CLOCK_GHZ[st]}
            if st in mix_ns:
                out[st]["frac_opcode_mix_weighted"] = per_simd * mix_ns[st] * 1e-9 / t
    return out


def scaling_model(ctx, ops, g_np, w2c_np, Ks_np, N, V, W, H, device):
    """What ONE GPU can say about the 1 -> 8 GPU curve of the north_star partition (views sharded, Gaussians replicated):
    this rank's whole step at C_local = V, V/2, V/4, V/8 views -- the compute a rank of a 1-, 2-, 4-, 8-GPU job does
    (projection, its backward and Adam stay O(N); blending, loss and the level-2 sort scale with the views) --, the
    library's own exchange on a one-rank communicator (what the collective's launch and the exact per-step sizing cost
    without any link traffic), and the link time of the 92 MB gradient buffer under SURVEY.md section 5's xGMI model
    (7 point-to-point links x 153.6 GB/s per GPU): ring all-reduce 2 (w-1)/w B / link, direct reduce-scatter +
    all-gather 2 B / (w link).  A MODEL, not a measurement: no multi-GPU node was available to the builder."""
    from starst3r_amd import dist as sdist
    LINK = 153.6e9
    grad_bytes = 23 * N * 4
    P0 = {k: torch.tensor(v, device=device) for k, v in g_np.items()}
    out = {"note": "single-GPU model of the view-sharded job (DESIGN.md section 5); predicted, NOT measured on >1 GPU",
           "link_model": "xGMI 7 links x 153.6 GB/s per GPU (SURVEY.md section 5)", "gradient_buffer_bytes": grad_bytes,
           "per_rank": {}}

    def time_steps(P, w2c, Ks, gt, n=10, warm=3):
        campos = ops.camera_positions(w2c)
        if id(gt) not in moms:
            moms[id(gt)] = ops.gt_moments(ctx, gt)
        ops.set_gt_moments(ctx, gt, moms[id(gt)])   # (like gs.run_3dgs_optim: once per training call)
        grads = torch.empty(23 * N, device=device); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        loss = torch.zeros(1, device=device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(warm + n):
            if it == warm:
                e0.record()
            ops.train_step(ctx, P, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                           loss, want_stats=(it == 0))
        e1.record(); torch.cuda.synchronize()
        try:
            ops.settle(ctx)
        except Exception:  # noqa: BLE001 -- a late capacity report of the last untimed step is of no interest here
            pass
        return e0.elapsed_time(e1) / n
    worlds = [w for w in (1, 2, 4, 8) if V % w == 0]
    setups = {}
    moms = {}
    for w in worlds:
        views = list(range(0, V, w))                      # rank 0's shard of a w-GPU job (bench.py: range(rank, V, world))
        w2c = torch.tensor(w2c_np[views], device=device); Ks = torch.tensor(Ks_np[views], device=device)
        gt = make_gt_images(ctx, ops, g_np, w2c, Ks, W, H, device)
        setups[w] = (w2c, Ks, gt)
        ms = time_steps({k: t.clone() for k, t in P0.items()}, w2c, Ks, gt)
        out["per_rank"][str(w)] = {"views_per_gpu": len(views), "step_ms_no_exchange": ms}
    # the exchange machinery with one rank: collective launch + exact sizing (one host round trip per step), no link traffic
    try:
        sdist.attach_native_comm(ctx)
        buf = torch.zeros(23 * N, device=device)
        from starst3r_amd import _lib
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r_ in range(6):
            if r_ == 1:
                e0.record()
            _lib.check(_lib.lib().st3r_grad_allreduce(ctx.handle, ops._stream(), ops._p(buf), buf.numel()))
        e1.record(); torch.cuda.synchronize()
        out["one_rank_allreduce_ms"] = e0.elapsed_time(e1) / 5
        for w in worlds:
            w2c, Ks, gt = setups[w]
            forms = {}
            for form in ops.EXCHANGE_FORMS:
                ops.set_exchange(ctx, form)
                forms[form] = time_steps({k: t.clone() for k, t in P0.items()}, w2c, Ks, gt, n=6, warm=2)
            out["per_rank"][str(w)]["step_ms_one_rank_communicator"] = forms
        ops.set_exchange(ctx, "allreduce")
        sdist.detach_native_comm(ctx)
    except Exception as e:  # noqa: BLE001 -- no RCCL the library can bind: the compute part of the model stands
        out["one_rank_allreduce_ms"] = None
        out["communicator_error"] = str(e)
    ops.set_gt_moments(ctx, None, None)
    adam_ms = 644.0 * N / 4.4e12 * 1e3    # k_adam streams 644 B per Gaussian at ~4.4 TB/s (stage_ms.adam)
    pred = {}
    for w in worlds:
        base = out["per_rank"][str(w)]
        comm_ms = (base.get("step_ms_one_rank_communicator") or {}).get("allreduce")
        # compute of a rank incl. the communicator's fixed costs when they could be measured
        local = comm_ms if comm_ms is not None else base["step_ms_no_exchange"]
        ring = 2.0 * (w - 1) / w * grad_bytes / LINK * 1e3 if w > 1 else 0.0
        direct = 2.0 * grad_bytes / (w * LINK) * 1e3 if w > 1 else 0.0
        pred[str(w)] = {
            "link_ms_ring_allreduce": ring, "link_ms_direct_rs_ag": direct,
            "iters_per_sec_allreduce_form_ring": 1e3 / (local + ring),
            # `direct` form (csrc/comm.hip, round 5): this library's own one-shot reduce-scatter + all-gather over HIP-IPC
            # peer windows -- every rank READS 1/w of the buffer from each peer over that peer's link, twice
            "iters_per_sec_direct_form": 1e3 / (local + direct - adam_ms * (1.0 - 1.0 / w)),
            # RCCL's reduce-scatter + all-gather (rs_ag form): whichever algorithm RCCL picks; a ring moves the same bytes
            # per link as the ring all-reduce
            "iters_per_sec_rs_ag_form_if_ring": 1e3 / (local + ring - adam_ms * (1.0 - 1.0 / w)),
            # layout 2 (Gaussians and views sharded): same blending / loss / sorts per rank, Adam on 1/w of the Gaussians,
            # two all-to-alls of 48-byte records over direct links instead of the gradient exchange
            "iters_per_sec_gaussian_sharded_direct": 1e3 / (local - adam_ms * (1.0 - 1.0 / w) +
                                                            (2.0 * 48 * (V // w) * N / w / LINK * 1e3 if w > 1 else 0.0)),
        }
    out["predicted"] = pred
    out["assumptions"] = ("exchange not overlapped with compute; ring = RCCL's default algorithm for all-reduce, "
                          "reduce-scatter and all-gather (which one RCCL 2.26 picks on this mesh is UNKNOWN: a one-rank "
                          "communicator logs no tuning decision, profiles/r5_rccl_one_rank_probe.md); direct = the "
                          "`direct` exchange form, whose algorithm is the library's own (peer reads over HIP IPC): "
                          "1/w of the buffer per link and phase BY CONSTRUCTION, its link rate and barrier cost "
                          "unmeasured; the piece-wise rows subtract the replicated Adam's share those forms do not run")
    return out


def config1_bench(ctx, ops, device, steps=60, warm=15):
    """BASELINE.json configs[1]'s training regime on one GPU -- 200 k Gaussians, 8 views of 512 x 384 (the size the
    reference's own main.py:80-81 runs) --, labelled, NOT the headline: the same single C call per iteration, ground-truth
    moments registered once like run_3dgs_optim does."""
    from st3r_synth import synth
    N, V, W, H = 200_000, 8, 512, 384
    g_np, w2c_np, Ks_np = synth.make_scene(N, V, W, H)
    P = {k: torch.tensor(v, device=device) for k, v in g_np.items()}
    P["shN"] = P["shN"][:, :4].contiguous()
    w2c, Ks = torch.tensor(w2c_np, device=device), torch.tensor(Ks_np, device=device)
    campos = ops.camera_positions(w2c)
    gt = make_gt_images(ctx, ops, g_np, w2c, Ks, W, H, device)
    mom = ops.gt_moments(ctx, gt)
    ops.set_gt_moments(ctx, gt, mom)
    grads = torch.empty(23 * N, device=device); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    loss = torch.zeros(warm + steps, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st0 = None
    try:
        for it in range(warm + steps):
            if it == warm:
                e0.record()
            st = ops.train_step(ctx, P, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                                loss[it:it + 1], want_stats=(it == 0))
            st0 = st0 or st
        e1.record(); torch.cuda.synchronize()
        ops.settle(ctx)
    finally:
        ops.set_gt_moments(ctx, None, None)
    ms = e0.elapsed_time(e1) / steps
    L = loss.cpu().numpy()
    return {"workload": f"BASELINE configs[1] regime: {N} gaussians, {V} views {W}x{H}, train only (labelled secondary "
                        "measurement, not the headline)", "ms_per_step": ms, "iters_per_sec": 1e3 / ms, "steps": steps,
            "warmup": warm, "n_isects_kept_first_step": (st0 or {}).get("n_isects"), "loss_first": float(L[0]),
            "loss_last": float(L[-1]), "seconds_for_7000_iterations_at_this_rate": 7.0 * ms}


def replica_checksum(P):
    """one int64 over the BITS of every parameter tensor: equal on every rank <=> the replicas are (all but certainly) identical"""
    tot = torch.zeros((), dtype=torch.int64, device=next(iter(P.values())).device)
    for k in sorted(P):
        b = P[k].contiguous().view(torch.int32).to(torch.int64)
        tot = tot + (b * (torch.arange(b.numel(), device=b.device, dtype=torch.int64).reshape(b.shape) % 1021 + 1)).sum()
    return tot


def matching_bench(device, with_cpu=True):
    """Path A: seeded nearest-neighbour query of fast_reciprocal_NNs (starster/reconstruct.py:97) at the
    reference's size: 3072 seeds against the 512x384 descriptors (D = 24) of the other image -- the only dense
    contraction of the system, bounded by the fp32 MFMA peak."""
    from starst3r_amd import matching, ops
    ctx = ops.get_context(device)
    H, W, D, n = 384, 512, 24, 3072
    gen = torch.Generator(device=device).manual_seed(0)
    A = torch.nn.functional.normalize(torch.randn(H * W, D, device=device, generator=gen), dim=1)
    B = torch.nn.functional.normalize(torch.randn(H * W, D, device=device, generator=gen), dim=1)
    # planted correspondences so the reciprocal iteration converges the way real descriptors do
    perm = torch.randperm(H * W, device=device, generator=gen)[: H * W // 3]
    B[perm] = torch.nn.functional.normalize(A[perm] + 0.05 * torch.randn(perm.numel(), D, device=device, generator=gen), dim=1)
    q = A[:n].contiguous()
    matching.nn_dot_argmax(ctx, q, B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        matching.nn_dot_argmax(ctx, q, B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tflops = 2.0 * n * H * W * D / (ms * 1e-3) / 1e12
    A3, B3 = A.reshape(H, W, D), B.reshape(H, W, D)
    matching.fast_reciprocal_NNs(A3, B3, 8, ret_xy=False, device=device); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        i1, _ = matching.fast_reciprocal_NNs(A3, B3, 8, ret_xy=False, device=device)
    torch.cuda.synchronize()
    recip_ms = (time.perf_counter() - t0) / 5 * 1e3
    cpu = None
    if with_cpu:   # the numpy restatement (oracle/nn_oracle.py: blocked fp32 matmul + arg-max) on the host cores
        from oracle import nn_oracle
        qn, bn = q.cpu().numpy(), B.cpu().numpy()
        t0 = time.perf_counter()
        nn_oracle.nn_dot(qn, bn, dtype=np.float32)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        cpu = {"value": cpu_ms, "unit": "ms per query call", "cores": os.cpu_count(), "kind": "port",
               "sample": "the same 3072 x 196608 x 24 query, numpy sgemm + argmax (BLAS threads = all cores)",
               "speedup": cpu_ms / ms}
    return {"query": f"{n} seeds x {H * W} descriptors, D={D}", "query_ms": ms, "cpu_baseline": cpu,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": 157.3, "unit": "TFLOP/s", "frac": tflops / 157.3,
                         # tools/probe/mfma_f32_peak.hip on this part: back-to-back v_mfma_f32_32x32x2_f32, random operands
                         "peak_measured": 153.4, "frac_of_measured": tflops / 153.4,
                         "note": "fp32 v_mfma_f32_32x32x2_f32; flops = 2*n*m*D, score matrix never written; "
                                 "time = arg-max kernel + per-query row resolution kernel"},
            "fast_reciprocal_NNs_ms": recip_ms, "matches": int(i1.numel()),
            "loop": "device resident (st3r_recip_nn), 10 reciprocal iterations, no host sync"}


def main():
    args = parse()
    # The ONE JSON line is the only thing that may reach stdout: libraries print there too (RCCL's version banner at
    # communicator creation, from C, buffered until exit) -- so file descriptor 1 points at stderr for the whole run and
    # the line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # ST3R_BENCH_EMULATE_RANKS=1 (tools/experiments/bench_emulated_ranks.sh; never for a reported number): every rank
    # uses cuda:0, torch.distributed runs on gloo and the library's communicator binds ST3R_RCCL_LIB (tests/fake_rccl) --
    # a crash test of the N > 1 code path of this file on a one-GPU box; the line says "emulated" in config.parallelism
    EMULATED = os.environ.get("ST3R_BENCH_EMULATE_RANKS") == "1"
    if EMULATED:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run the RCCL path is taken even with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if EMULATED:
            assert os.environ.get("ST3R_RCCL_LIB"), "emulated ranks need ST3R_RCCL_LIB (tests/_build/libfake_rccl.so)"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from starst3r_amd import ops
    from st3r_synth import synth
    ctx = ops.get_context(device)

    N, W, H = args.gaussians, args.width, args.height
    assert args.views % world == 0, "views must divide evenly over the GPUs"
    g_np, w2c_np, Ks_np = synth.make_scene(N, args.views, W, H)
    C_local = args.views // world
    # N GPUs: the headline layout is the north_star partition -- views sharded, Gaussians
    # replicated, one sum-all-reduce of the [23N] gradient buffer per iteration, issued by the library itself
    # (st3r_gs_train_step over its own RCCL communicator).  --multi-gpu gaussian-sharded is the labelled alternative
    # (two all-to-alls of splat records, nothing replicated).
    # ST3R_BENCH_FREEZE=1 (tools/experiments/abl.sh): no optimizer step, so that kernel variants with deliberately broken
    # gradients are all timed on the same scene; never set for a reported number
    FREEZE = os.environ.get("ST3R_BENCH_FREEZE") == "1"
    mode = args.multi_gpu
    total = args.warmup + args.steps
    DRIFT_TO = 200   # SURVEY 8(d) asks for >= 200 steady-state iterations; the driver fixes --steps 20
    want_drift = world == 1 and not FREEZE and not args.no_drift and args.steps < DRIFT_TO and mode != "gaussian-sharded"
    losses = torch.zeros(max(total, args.warmup + DRIFT_TO) if want_drift else total, device=device)
    stats = {}
    gtm_ms = None
    if mode == "gaussian-sharded":
        from starst3r_amd import dist as sdist
        views = sdist.shard_views_contiguous(args.views, rank, world)
        lo, hi = sdist.shard_gaussians(N, rank, world)
        w2c_all = torch.tensor(w2c_np, device=device); Ks_all = torch.tensor(Ks_np, device=device)
        w2c = w2c_all[views].contiguous(); Ks = Ks_all[views].contiguous()
        gt = make_gt_images(ctx, ops, g_np, w2c, Ks, W, H, device)
        P = {k: torch.tensor(np.ascontiguousarray(v[lo:hi]), device=device) for k, v in g_np.items()}
        trainer = sdist.ShardedTrainer(ctx, P, N, w2c_all, Ks_all, gt, W, H, rank, world)

        def step(it):
            return trainer.step(losses[it:it + 1])
    else:
        views = list(range(rank, args.views, world))  # this rank's views
        P = {k: torch.tensor(v, device=device) for k, v in g_np.items()}
        # like gs.run_3dgs_optim: the loop trains the four SH rows the kernels use as a compact [N, 4, 3] tensor (sh_stride 12;
        # rows 4..23 of the reference's [N, 24, 3] `shN` never change) -- ST3R_BENCH_COMPACT_SH=0: the 288-byte-stride rows in place
        if os.environ.get("ST3R_BENCH_COMPACT_SH", "1") != "0":
            P["shN"] = P["shN"][:, :4].contiguous()
        w2c = torch.tensor(w2c_np[views], device=device)
        Ks = torch.tensor(Ks_np[views], device=device)
        campos = ops.camera_positions(w2c)
        gt = make_gt_images(ctx, ops, g_np, w2c, Ks, W, H, device)
        grads = torch.empty(23 * N, device=device)
        m = torch.zeros_like(grads); v = torch.zeros_like(grads)
        # like gs.run_3dgs_optim: SSIM's two ground-truth moments conv(gt), conv(gt^2) are computed ONCE per training call
        # (the images do not change inside it, starster/gs.py:149-152) and registered with the ctx; the timed steps read
        # them.  The pre-pass is timed here and reported (config.gt_moments_once_per_call_ms).  ST3R_BENCH_GT_MOMENTS=0:
        # every step convolves the ground truth itself, as up to round 5
        if os.environ.get("ST3R_BENCH_GT_MOMENTS", "1") != "0":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gt_moments(ctx, gt); e0.record()
            gt_mom = ops.gt_moments(ctx, gt)
            e1.record(); torch.cuda.synchronize()
            gtm_ms = e0.elapsed_time(e1)
            ops.set_gt_moments(ctx, gt, gt_mom)

        native_comm = True
        if dist is not None:   # one process per GPU (under torch.distributed.run also with a single rank: same code path)
            from starst3r_amd import dist as sdist
            try:
                sdist.attach_native_comm(ctx)      # torch.distributed only ships the 128-byte RCCL id
                ok = torch.ones(1, device=device)
            except Exception as e:  # noqa: BLE001 -- e.g. no RCCL the library can bind: reported, not fatal
                print(f"[bench] rank {rank}: library communicator unavailable ({e})", file=sys.stderr)
                ok = torch.zeros(1, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # all ranks take the same path
            native_comm = bool(ok.item() > 0)
            if not native_comm and getattr(ctx, "native_comm", False):
                sdist.detach_native_comm(ctx)

        def step(it):
            # statistics (which need the record count on the host, i.e. a device synchronisation) only for the first
            # and the last step; every other step runs without a host round trip, like Scene.run_3dgs_optim
            stats = it == 0 or it == total - 1
            if FREEZE:
                return ops.train_fwd_bwd(ctx, P, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, losses[it:it + 1],
                                         want_stats=stats)
            if not native_comm:   # the exchange through the host framework's process group (torch.distributed = RCCL)
                st_ = ops.train_fwd_bwd(ctx, P, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, losses[it:it + 1],
                                        want_stats=stats)
                dist.all_reduce(grads, op=dist.ReduceOp.SUM)
                ops.adam_step(ctx, P, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1)
                return st_
            # the whole iteration is ONE C call: fwd/bwd -> st3r_grad_allreduce (RCCL, no-op for one rank) -> Adam
            return ops.train_step(ctx, P, w2c, Ks, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999,
                                  1e-8, it + 1, losses[it:it + 1], want_stats=stats)

    def psnr_local():
        """mean PSNR of this rank's views against their GT images (renders outside the timed region)"""
        if mode == "gaussian-sharded":
            return None
        with torch.no_grad():
            rgb, _, _ = ops.render(ctx, P, w2c, Ks, ops.camera_positions(w2c), W, H)
            mse = ((rgb.clamp(0, 1) - gt) ** 2).reshape(rgb.shape[0], -1).mean(1)
            return float((-10.0 * torch.log10(mse)).mean())
    psnr_before = psnr_local()

    # Stage timing costs two HIP events per stage and step (~0.15 ms per step for all eleven): the warm-up steps time
    # every stage to find the dominant one; in the timed region ONE stage is timed per step (two events, live, on the
    # launch stream): the dominant stage on every other step, the ten others in turn on the steps between.
    ops.set_profiling(ctx, args.warmup > 0)
    ops.stage_ms(ctx)  # reset
    for it in range(args.warmup):
        stats = step(it) or stats
    warm_stage = ops.stage_ms(ctx) if args.warmup > 0 else {}
    warm_ms = {k: ms / n for k, (ms, n) in warm_stage.items() if n > 0}
    dom_warm = max(warm_ms, key=warm_ms.get) if warm_ms else None
    others = [k for k in ops.STAGES if k != dom_warm]
    sched = [dom_warm if (i % 2 == 0 or not others) else others[(i // 2) % len(others)] for i in range(args.steps)]
    ops.set_profiling(ctx, dom_warm is not None, only=dom_warm)
    ops.stage_ms(ctx)  # reset
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per step on the launch stream (a few microseconds each): the K timed steps stay ONE uninterrupted
    # region, and its five windows of K/5 steps are read back from the events afterwards
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for it in range(args.warmup, total):
        if dom_warm is not None:
            ops.set_profiling(ctx, True, only=sched[it - args.warmup])   # (a mask in the ctx: no device work)
        stats = step(it) or stats
        marks[it - args.warmup + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_win = 5 if args.steps >= 5 else 1
    edges = [round(k * args.steps / n_win) for k in range(n_win + 1)]
    win_ms = [marks[edges[k]].elapsed_time(marks[edges[k + 1]]) / (edges[k + 1] - edges[k]) for k in range(n_win)]
    stage = ops.stage_ms(ctx)
    ops.set_profiling(ctx, False)
    psnr_after = psnr_local()
    # The reference trains RAW scales: the Gaussians grow and the step gets slower with the step count.
    # Untimed continuation of the same run to 200 steps: the step time at steps 100-120 and 180-200 next to the timed one.
    drift = None
    if want_drift:
        bounds = [b for b in (100, 120, 180, 200) if b > args.steps]
        ev = {}
        for it in range(total, args.warmup + DRIFT_TO):
            k = it - args.warmup
            if k in bounds:
                ev[k] = torch.cuda.Event(enable_timing=True); ev[k].record()
            step(it)
        ev[DRIFT_TO] = torch.cuda.Event(enable_timing=True); ev[DRIFT_TO].record()
        torch.cuda.synchronize()
        drift = {f"steps_0_{args.steps}": dt / args.steps * 1e3}
        for a, b in ((100, 120), (180, 200)):
            if a in ev and b in ev:
                drift[f"steps_{a}_{b}"] = ev[a].elapsed_time(ev[b]) / (b - a)
    # What the once-per-call ground-truth moments are worth per step, measured: blocks of 5 more steps alternately WITH the
    # registered moments (as in the timed region, as in gs.run_3dgs_optim) and WITHOUT (every step convolves the ground truth
    # itself, like torchmetrics inside the reference's compute_loss) -- so that the line also carries the rate a reader gets
    # who counts the moments as per-iteration work
    gtm_delta_ms = None
    if gtm_ms is not None and world == 1 and not FREEZE and mode != "gaussian-sharded":
        it_x = max(losses.numel() - 2, 1)   # (not the first / last index: those ask for statistics, i.e. the synchronous path)
        acc = {True: 0.0, False: 0.0}
        for blk in range(6):
            use = blk % 2 == 0
            ops.set_gt_moments(ctx, gt if use else None, gt_mom if use else None)
            step(it_x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                step(it_x)
            e1.record(); torch.cuda.synchronize()
            acc[use] += e0.elapsed_time(e1) / 5
        ops.set_gt_moments(ctx, gt, gt_mom)
        gtm_delta_ms = (acc[False] - acc[True]) / 3
    # per-rank view of the same timed region, and the exchange on its own (N > 1)
    my_ms = marks[0].elapsed_time(marks[-1]) / args.steps
    per_rank_ms, exch_ms = [my_ms], None
    if dist is not None:
        t_all = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([my_ms], device=device, dtype=torch.float64))
        per_rank_ms = [float(x.item()) for x in t_all]
        if mode != "gaussian-sharded":
            from starst3r_amd import _lib
            scratch = torch.zeros(23 * N, device=device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            for r_ in range(reps + 1):
                if r_ == 1:
                    dist.barrier(); torch.cuda.synchronize(); e0.record()
                if native_comm:
                    _lib.check(_lib.lib().st3r_grad_allreduce(ctx.handle, ops._stream(), ops._p(scratch), scratch.numel()))
                else:
                    dist.all_reduce(scratch)
            e1.record(); torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exch_ms = float(t.item())
    # N > 1 self-diagnosis: how many ranks the library's communicator really has, and whether the
    # replicas are still identical after the timed steps (a wrong exchange shows as a difference, not as a slow step)
    n_ranks_seen, replicas_identical = 1, None
    if mode != "gaussian-sharded":
        from starst3r_amd import _lib as _l
        import ctypes
        ws, rk = ctypes.c_int(0), ctypes.c_int(0)
        _l.check(_l.lib().st3r_comm_world(ctx.handle, ctypes.byref(ws), ctypes.byref(rk)))
        n_ranks_seen = int(ws.value)
        if dist is not None:
            mine = replica_checksum(P).reshape(1)
            allc = [torch.zeros_like(mine) for _ in range(world)]
            if EMULATED:
                allc = [a.cpu() for a in allc]; dist.all_gather(allc, mine.cpu())
            else:
                dist.all_gather(allc, mine)
            replicas_identical = all(int(a.item()) == int(allc[0].item()) for a in allc)
    if mode == "gaussian-sharded":   # counts of the own Gaussians over all views ~ those of the own views over all Gaussians
        stats["n_visible"] = int(trainer.reg[2].item()); stats["n_isects_ref"] = int(trainer.reg[3].item())
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        lt = losses.clone(); dist.all_reduce(lt); losses = lt
    L = losses.cpu().numpy()

    if rank == 0:
        # I = tile intersections of the reference algorithm (gsplat's 3-sigma squares): the unit the algorithmic-byte
        # formula is written in.  The fused path sorts/blends only the I_kept of them whose alpha >= 1/255 box
        # reaches the tile (the dropped ones fail the alpha test on all 256 pixels).
        V, I, I_kept = stats["n_visible"], stats["n_isects_ref"], stats["n_isects"]
        P_px = C_local * H * W
        tw, th = ops.tile_grid(W, H)
        keybits = 32 + (tw * th).bit_length() + C_local.bit_length()           # gsplat's single key (reference)
        key1_bytes = 4 if C_local <= 8 else 8                                   # level-1 (camera | depth) key
        key1_bits = (29 if C_local <= 8 else 32) + max(C_local - 1, 0).bit_length()
        # round 6: <= 8 views sort per camera segment on depth codes biased by the smallest one of the call, in as many 8-bit
        # passes as that range needs -- decided on the device (control word 10 of the fused steps); the bytes are those of
        # the passes that RAN
        level1_passes_run = None
        if C_local <= 8 and mode != "gaussian-sharded":
            level1_passes_run = int(ops.peek(ctx, 10, 16)[10].item())
            if 1 <= level1_passes_run <= 4:
                key1_bits = 8 * level1_passes_run
        key2_bits = max(C_local * tw * th - 1, 1).bit_length()                  # level-2 (camera, tile) key
        ab = algorithmic_bytes(N, C_local, V, I_kept, P_px, C_local * tw * th, key1_bits, key1_bytes, key2_bits)
        ab_ref = algorithmic_bytes_reference(N, V, I, P_px, C_local * tw * th, keybits)
        per_stage = dict(warm_ms)                      # fall-back: warm-up steps
        timed_samples = {k: n for k, (ms, n) in stage.items() if n > 0}
        per_stage.update({k: ms / n for k, (ms, n) in stage.items() if n > 0})   # timed region, one stage per step
        dom = dom_warm if dom_warm is not None else max(per_stage, key=per_stage.get)
        dom_ms = per_stage[dom]
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        ms_per_step = dt / args.steps * 1e3
        iter_bytes = sum(ab.values())
        traffic, traffic_src = pmc_traffic(dom, N, args.views, W, H, world)
        win_sorted = sorted(win_ms)
        out = {
            "metric": "3DGS train iters/sec @ 1M Gaussians, 8x1080p views",
            "value": args.steps / dt, "unit": "iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"SYNTH-1M (BASELINE configs[2]): {N} gaussians, {args.views} views {W}x{H}, train only; "
                            f"{C_local} views/GPU, gaussians " + ("sharded" if mode == "gaussian-sharded" else "replicated"),
                "gaussians": N, "views": args.views, "width": W, "height": H, "views_per_gpu": C_local,
                "parallelism": (f"gaussians+views sharded x{world} (2 all-to-all of splat records / iteration)"
                                if mode == "gaussian-sharded"
                                else f"view-dp{world} (st3r_grad_allreduce of the [23N] gradients inside st3r_gs_train_step)"
                                if native_comm else
                                f"view-dp{world} (torch.distributed all_reduce of the [23N] gradients between "
                                f"st3r_gs_train_fwd_bwd and st3r_adam_step)"),
                **({"emulated_ranks": "ALL ranks share cuda:0 over a shared-memory RCCL stand-in: a crash test of the "
                                      "N > 1 path, not a measurement"} if EMULATED else {}),
                "n_visible_pairs": V, "n_isects_reference_algorithm": I, "n_isects_kept_after_exact_culling": I_kept,
                "sort_key_bits": {"reference_single_key": keybits, "level1": key1_bits, "level2": key2_bits},
                "level1_sort_passes_run": level1_passes_run,
                "mean_tiles_per_visible_gaussian": (I_kept / V) if V else 0.0,
                "mean_records_per_tile": I_kept / (C_local * tw * th),
                # loss of the first step and of the LAST TIMED step (step warmup + steps), the step psnr_db_after is taken
                # at; the loss at the end of the untimed continuation (windows.drift_ms_per_step) has its own key
                "loss_first": float(L[0]), "loss_last": float(L[total - 1]), "loss_last_step": total,
                **({"loss_after_drift_continuation": float(L[-1]), "drift_continuation_to_step": int(len(L))}
                   if drift else {}),
                # training views of rank 0 against their GT, before the first and after the last of the warmup + timed steps
                "psnr_db_before": psnr_before, "psnr_db_after": psnr_after,
                # SSIM's ground-truth moments: computed once per training call outside the timed steps (None: every step
                # convolves the ground truth itself)
                "gt_moments_once_per_call_ms": gtm_ms,
                # measured after the timed region (alternating blocks of steps with / without the registered moments): what a
                # step costs more when it convolves the ground truth itself, and the headline rate with that added back
                "gt_moments_recompute_cost_ms_per_step": gtm_delta_ms,
                "iters_per_sec_if_gt_moments_were_recomputed_every_step":
                    (1e3 / (ms_per_step + gtm_delta_ms)) if gtm_delta_ms is not None else None,
                **({"iters_per_sec_steps_180_200": 1e3 / drift["steps_180_200"],
                    "ms_per_step_steps_180_200": drift["steps_180_200"]} if drift and "steps_180_200" in drift else {}),
            },
            # five windows of steps/5 consecutive steps inside the one timed region (HIP events on the launch stream)
            "windows": {"ms_per_step": win_ms, "median_ms_per_step": win_sorted[len(win_sorted) // 2],
                        "median_iters_per_sec": 1e3 / win_sorted[len(win_sorted) // 2],
                        "spread_rel": (win_sorted[-1] - win_sorted[0]) / win_sorted[len(win_sorted) // 2],
                        # ms per step of the same run continued (untimed) to 200 steps: raw scales grow under training
                        "drift_ms_per_step": drift},
            # every rank's own ms per step over the timed region (HIP events on its launch stream) and, for N > 1, one
            # exchange of the [23N] gradient buffer on its own (max over ranks): step - exchange ~ what a rank computes
            "per_rank": {"ms_per_step": per_rank_ms, "exchange_ms_isolated": exch_ms,
                         # (the whole step under the other exchange forms -- ranges, rs_ag -- is timed AFTER this line is
                         # out and reported on stderr / gpurun_out/exchange_forms_n<N>.json: those forms have never run
                         # with more than one rank, and a hang there must not cost the measurement)
                         "exchange": ops.get_exchange(ctx) if (dist is not None and native_comm) else
                                     ("allreduce (torch.distributed)" if dist is not None else "none")},
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": ab[dom], "launch_ms": dom_ms,
                "formula": "SURVEY 8(d) bytes per unit x executed units (records = n_isects_kept) / mean launch ms (HIP events)",
                "whole_iter": {"algorithmic_bytes": iter_bytes,
                               "achieved": iter_bytes / (ms_per_step * 1e-3) / 1e9,
                               "frac": iter_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                # the same kernel time priced with the REFERENCE algorithm's bytes (gsplat's 3-sigma squares, 64-bit
                # single-key sort): how much of the reference's traffic per second the fused path retires -- not
                # achieved bandwidth
                "vs_reference_algorithm": {
                    "kernel_bytes": ab_ref.get(dom), "kernel_equiv_GBps": (ab_ref[dom] / (dom_ms * 1e-3) / 1e9) if dom in ab_ref and dom_ms > 0 else None,
                    "whole_iter_bytes": sum(ab_ref.values()),
                    "whole_iter_equiv_GBps": sum(ab_ref.values()) / (ms_per_step * 1e-3) / 1e9},
                # Since round 5 the per-pair sums of the backward's (record, tile) slots are taken inside the projection backward
                # (k_gather_vtile's 0.25 ms left the blend_bwd stage, project_bwd grew by 0.12): the two backward stages together,
                # comparable across rounds (round 4: 3.97 GB / 2.47 ms = 0.20)
                "backward_stages_combined": (lambda b, t: {"stages": "blend_bwd + project_bwd", "algorithmic_bytes": b, "ms": t,
                                                            "achieved": b / (t * 1e-3) / 1e9 if t > 0 else None,
                                                            "frac": b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS if t > 0 else None})(
                    ab["blend_bwd"] + ab["project_bwd"], per_stage.get("blend_bwd", 0.0) + per_stage.get("project_bwd", 0.0)),
                "algorithmic_bytes_by_stage": ab,
                "stage_ms": per_stage,
                "stage_ms_samples_in_timed_region": timed_samples,
                "stage_ms_source": f"HIP events in the timed region, one stage per step ({dom} every other step)",
            },
        }
        # the same numbers once more as FLAT scalars (the driver's record keeps the scalars of this object and drops nested
        # ones): ms per stage, the whole iteration against the roofline
        out["roofline"].update({f"ms_{k}": v for k, v in per_stage.items()})
        out["roofline"]["whole_iter_GBps"] = out["roofline"]["whole_iter"]["achieved"]
        out["roofline"]["whole_iter_frac"] = out["roofline"]["whole_iter"]["frac"]
        out["roofline"]["backward_stages_frac"] = out["roofline"]["backward_stages_combined"]["frac"]
        if drift and "steps_180_200" in drift:   # SURVEY 8(d)'s >= 200-step regime next to the driver's 20-step headline
            out["value_steps_180_200"] = 1e3 / drift["steps_180_200"]
        vi = valu_issue(per_stage, N, args.views, W, H, world)
        out["roofline"]["valu_issue"] = vi
        # the bound the blend kernels actually hit, as flat scalars: fraction of the 2-cycle VALU issue
        # rate at the 2.4 GHz the guide quotes (None until profiles/pmc_traffic.json belongs to these kernel sources)
        for st_ in ("blend_fwd", "blend_bwd"):
            out["roofline"][f"valu_issue_frac_{st_}"] = (vi or {}).get(st_, {}).get("frac_at_2_cycles")
        # DESIGN.md section 7, "floor": the VALU slots this formulation cannot avoid (forward: 22 per (record, wave) trip of
        # the cell lists; backward: 28 per trip + 60 per four records) x the trips SYNTH-1M takes x 2 cycles at the 2.1 GHz
        # the kernels hold = 0.48 + 1.10 ms, plus the 1.75 ms every other stage takes today = 3.3 - 3.4 ms: the 3.33 ms of
        # 300 it/s are not reachable by scheduling these loops better, only by evaluating fewer (record, pixel) pairs
        out["roofline"]["blend_formulation_floor_ms"] = {"blend_fwd": 0.48, "blend_bwd": 1.10,
                                                          "all_other_stages_today": sum(v_ for k_, v_ in per_stage.items() if k_ not in ("blend_fwd", "blend_bwd"))}
        out["roofline"]["target_reachable_with_this_formulation"] = False
        # flat N > 1 diagnostics
        out["n_ranks_seen"] = n_ranks_seen
        out["replicas_identical_after_timed_steps"] = replicas_identical
        out["exchange_ms_isolated"] = exch_ms
        out["compute_only_ms_per_step"] = (ms_per_step - exch_ms) if exch_ms is not None else ms_per_step
        if world == 1 and not FREEZE and not args.train_only and not args.no_config1 and mode != "gaussian-sharded":
            ops.set_gt_moments(ctx, None, None)
            out["config_1"] = config1_bench(ctx, ops, device)
            out["config_1_iters_per_sec"] = out["config_1"]["iters_per_sec"]
        if world == 1 and not FREEZE and not args.no_scaling_model and mode != "gaussian-sharded":
            out["scaling_model"] = scaling_model(ctx, ops, g_np, w2c_np, Ks_np, N, args.views, W, H, device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.train_only:
            out["align"] = align_bench(device, with_cpu=not args.no_cpu_baseline)
            # the synthetic condensed problems at 2 / 8 / 32 views (HIP seconds for 500+200 iterations)
            # (the CPU port beside it at all three sizes; 2 and 32 views on a shorter sample, 50+20 iterations x 10)
            with_cpu = not args.no_cpu_baseline
            others = {c: align_bench(device, with_cpu=with_cpu, views=c, cpu_iters=(50, 20)) for c in (2, 32)}
            out["align"]["hip_seconds_by_views"] = {str(c): r["hip_seconds"] for c, r in others.items()}
            out["align"]["hip_seconds_by_views"]["8"] = out["align"]["hip_seconds"]
            if with_cpu:
                out["align"]["cpu_port_seconds_by_views"] = {str(c): r["cpu_port_seconds"] for c, r in others.items()}
                out["align"]["cpu_port_seconds_by_views"]["8"] = out["align"]["cpu_port_seconds"]
            out["matching"] = matching_bench(device, with_cpu=not args.no_cpu_baseline)
            out["condense"] = condense_bench(device, with_cpu=not args.no_cpu_baseline)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    # ---- after the line: the whole step under each exchange form of st3r_gs_train_step (a few extra steps each; rs_ag
    # leaves the Adam moments sharded, which no longer matters here).  Every rank arms a watchdog first: the forms other
    # than the plain all-reduce have only ever run with one rank, and a hung collective cannot be interrupted in-process.
    if dist is not None and mode != "gaussian-sharded" and native_comm and not FREEZE:
        import threading
        wd = threading.Timer(120.0, lambda: (sys.stderr.write("bench.py: exchange-form timing timed out\n"), os._exit(0)))
        wd.daemon = True; wd.start()
        forms_ms = {}
        keep = ops.get_exchange(ctx)
        it_x = total - 1
        # (`direct` last: it is the one form that has never met real peer-to-peer links -- set-up over HIP IPC, device-side
        # barriers --; an error there is recorded, not raised; the record of the RCCL forms is written BEFORE it is tried,
        # and its barriers give up after 2 s each so that a window that maps but does not synchronise cannot eat the watchdog)
        def record():
            if rank == 0:
                rec = {"n_gpus": world, "exchange_forms_ms_per_step": forms_ms}
                print("exchange forms: " + json.dumps(rec), file=sys.stderr, flush=True)
                try:
                    os.makedirs("gpurun_out", exist_ok=True)
                    with open(os.path.join("gpurun_out", f"exchange_forms_n{world}.json"), "w") as f:
                        json.dump(rec, f)
                except OSError:
                    pass
        os.environ.setdefault("ST3R_XBAR_TIMEOUT_MS", "2000")
        for form in ops.EXCHANGE_FORMS:
            if form == "direct":
                record()
            try:
                ops.set_exchange(ctx, form)
                step(it_x)                                   # warm-up of the form (streams, staging buffers, windows)
                if form == "direct":
                    ops.settle(ctx)                          # a barrier that gave up shows here, before anything is timed
                dist.barrier(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    step(it_x)
                e1.record(); torch.cuda.synchronize()
                t = torch.tensor([e0.elapsed_time(e1) / 5], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                forms_ms[form] = float(t.item())
                # the replicas after this form's six steps: identical or the form is WRONG on this node
                mine = replica_checksum(P).reshape(1)
                lo, hi = mine.clone(), mine.clone()
                if EMULATED:
                    lo, hi = lo.cpu(), hi.cpu()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                forms_ms[form + "_replicas_identical"] = bool(int(lo.item()) == int(hi.item()))
            except Exception as e:  # noqa: BLE001 -- a form that fails on this node is a finding, not a crash of the bench
                forms_ms[form] = f"failed on rank {rank}: {e}"[:200]
                break
        try:
            ops.set_exchange(ctx, keep)
        except Exception:  # noqa: BLE001
            pass
        wd.cancel()
        record()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
