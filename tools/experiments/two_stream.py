"""Experiment (round 4): do the latency-bound front-end kernels of one half of the views hide under the VALU-bound blend
kernels of the other half?  Two contexts (own arenas), each with 4 of SYNTH-1M's 8 views, stepped (forward + backward, no
Adam) on two streams -- against one context with all 8 views on one stream.  python tools/experiments/two_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from starst3r_amd import ops
from st3r_synth import synth

dev = torch.device("cuda:0")
N, V, W, H = 1_000_000, 8, 1920, 1080
g, w2c_np, Ks_np = synth.make_scene(N, V, W, H)
P = {k: torch.tensor(v, device=dev) for k, v in g.items()}
ctx_all = ops.Context("cuda:0")


def setup(ctx, views):
    w2c = torch.tensor(w2c_np[views], device=dev); Ks = torch.tensor(Ks_np[views], device=dev)
    gt = torch.rand(len(views), H, W, 3, device=dev)
    return dict(ctx=ctx, w2c=w2c, Ks=Ks, campos=ops.camera_positions(w2c), gt=gt, grads=torch.empty(23 * N, device=dev),
                loss=torch.zeros(1, device=dev))


def step(S, stats=False):
    ops.train_fwd_bwd(S["ctx"], P, S["w2c"], S["Ks"], S["campos"], S["gt"], W, H, 0.2, 0.01, 0.01, S["grads"], S["loss"],
                      want_stats=stats)


def timed(fn, n=20, warm=5):
    for i in range(warm):
        fn(i == 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

A = setup(ctx_all, list(range(8)))
t_all = timed(lambda st: step(A, st))
ctx1, ctx2 = ops.Context("cuda:0"), ops.Context("cuda:0")
B1, B2 = setup(ctx1, [0, 1, 2, 3]), setup(ctx2, [4, 5, 6, 7])
t_seq = timed(lambda st: (step(B1, st), step(B2, st)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(st):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        step(B1, st)
    with torch.cuda.stream(s2):
        step(B2, st)
    cur.wait_stream(s1); cur.wait_stream(s2)
t_par = timed(both)
print(f"8 views, one stream: {t_all:.3f} ms | 4 + 4 views back to back: {t_seq:.3f} ms | 4 + 4 views on two streams: {t_par:.3f} ms")

# free-running streams, the second one offset by ~half a front end + blend forward, no join per iteration
def free_run(offset_ms, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        if offset_ms > 0:
            torch.cuda._sleep(int(offset_ms * 1e-3 * 2.1e9))
    for _ in range(n):
        with torch.cuda.stream(s1):
            step(B1)
        with torch.cuda.stream(s2):
            step(B2)
    cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for off in (0.0, 0.6, 1.2, 1.8):
    free_run(off, 5)
    print(f"free-running, stream 2 offset by {off} ms: {free_run(off):.3f} ms per pair of half steps")
