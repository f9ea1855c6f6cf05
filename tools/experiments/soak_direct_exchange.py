"""Soak of the `direct` exchange form with emulated ranks on ONE GPU: W processes share cuda:0, the library's communicator
runs over tests/fake_rccl (only the IPC-handle exchange goes through it), every step moves the gradients through real
same-device HIP-IPC mappings and two device-side barriers.  Replicas are compared bit for bit every 100 steps.
    python tools/experiments/soak_direct_exchange.py [world=4] [steps=2000]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ST3R_EXCHANGE="direct",
                      ST3R_RCCL_LIB=os.path.join(ROOT, "tests", "_build", "libfake_rccl.so"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from starst3r_amd import dist as sdist, ops
    from st3r_synth import synth
    dev = torch.device("cuda:0")
    ctx = ops.get_context(dev)
    N, V, W, H = 30011, 2 * world, 320, 240          # 23 N not divisible by the world sizes: a tail everybody updates
    g, w2c, Ks = synth.make_scene(N, V, W, H, seed=5, scale_lo=0.004, scale_hi=0.03)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    vm_all, K_all = T(w2c), T(Ks)
    Q = {k: T(v) for k, v in synth.perturb_for_gt(g).items()}
    gt_all, _, _ = ops.render(ctx, Q, vm_all, K_all, ops.camera_positions(vm_all), W, H)
    views = sdist.shard_views(V, rank, world)
    vm, K, gt = vm_all[views].contiguous(), K_all[views].contiguous(), gt_all[views].clamp(0, 1).contiguous()
    campos = ops.camera_positions(vm)
    P = {k: T(v) for k, v in g.items()}
    sdist.attach_native_comm(ctx)
    assert ops.get_exchange(ctx) == "direct"
    grads = torch.empty(23 * N, device=dev); m = torch.zeros_like(grads); v = torch.zeros_like(grads)
    losses = torch.zeros(steps, device=dev)
    t0 = time.time()
    for it in range(steps):
        ops.train_step(ctx, P, vm, K, campos, gt, W, H, 0.2, 0.01, 0.01, grads, m, v, 1e-3, 0.9, 0.999, 1e-8, it + 1,
                       losses[it:it + 1])
        if (it + 1) % 100 == 0:
            torch.cuda.synchronize()
            for k, t in P.items():
                ref = t.detach().cpu().clone()
                torch.distributed.broadcast(ref, src=0)
                assert torch.equal(ref, t.detach().cpu()), (rank, it, k)
    torch.cuda.synchronize()
    L = losses.cpu()
    torch.distributed.all_reduce(L)
    if rank == 0:
        print(f"world {world} steps {steps} sec {time.time() - t0:.1f} finite {bool(torch.isfinite(L).all())} "
              f"loss {float(L[0]):.5f} -> {float(L[-1]):.5f}; replicas bit-identical at every check", flush=True)
    sdist.detach_native_comm(ctx)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    mp.spawn(worker, args=(world, 29577, steps), nprocs=world, join=True)
