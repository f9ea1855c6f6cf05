#!/usr/bin/env python
"""What does RCCL itself say about the algorithm / protocol of the three collectives of the gradient exchange?
One rank (the builder's box has one GPU; RCCL refuses two ranks on one device), NCCL_DEBUG=INFO with the TUNING / COLL /
GRAPH subsystems, 92 MB float32 (the [23 N] buffer at 1 M Gaussians).  Output: gpurun_out/rccl_tuning_probe.log
Run on the GPU box:  python tools/probe/rccl_tuning_probe.py
"""
import os
import subprocess
import sys

OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
os.makedirs(OUT, exist_ok=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    n = 23 * 1_000_000
    x = torch.ones(n, device="cuda:0"); y = torch.empty(n, device="cuda:0")
    for _ in range(2):
        dist.all_reduce(x)
        dist.reduce_scatter_tensor(y, x)
        dist.all_gather_into_tensor(x, y)
    torch.cuda.synchronize()
    print("RCCL version (torch):", torch.cuda.nccl.version())
    dist.destroy_process_group()
    sys.exit(0)

env = dict(os.environ, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,ENV,GRAPH,TUNING,COLL", HSA_ENABLE_IPC_MODE_LEGACY="0")
with open(os.path.join(OUT, "rccl_tuning_probe.log"), "w") as f:
    for extra in ({}, {"NCCL_ALGO": "Ring"}, {"NCCL_ALGO": "Tree"}, {"NCCL_PROTO": "Simple"}):
        f.write(f"===== extra env: {extra}\n"); f.flush()
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(env, **extra), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
        f.write(r.stdout[-20000:]); f.write(f"\n[rc {r.returncode}]\n")
    r = subprocess.run("rocm-smi --showtopo 2>&1 | head -60; ls /opt/rocm/lib/librccl* ; strings /opt/rocm/lib/librccl.so | "
                       "grep -iE 'RCCL_(DIRECT|ENABLE|FORCE|MSCCL|P2P|LL128|PIVOT)[A-Z_]*' | sort -u | head -80",
                       shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    f.write("===== topology / library / env knobs compiled into librccl.so\n" + r.stdout)
print("wrote", os.path.join(OUT, "rccl_tuning_probe.log"))
