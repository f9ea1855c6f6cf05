"""Repeated sorts of the shapes the train step uses (both tile shapes of radix_sort.hip), each checked against
torch's stable sort: a race in the chained scan would show up as a rare mismatch.  python tools/sort_stress.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from starst3r_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = ops.get_context(torch.device("cuda:0"))
g = torch.Generator(device="cuda:0").manual_seed(1)
bad = 0
for rep in range(reps):
    for n, bits in ((8_000_000, 32), (26_000_000, 16), (1_000_000, 32), (3_250_000, 16), (1_638_401, 29)):
        keys = torch.randint(0, 2 ** min(bits, 31), (n,), device="cuda:0", generator=g, dtype=torch.int64).to(torch.int32)
        if bits == 16:   # tile-key like: long runs
            keys = (torch.arange(n, device="cuda:0") // 37 % 65280).to(torch.int32)[torch.randperm(n, device="cuda:0", generator=g)]
        vals = torch.arange(n, device="cuda:0", dtype=torch.int32)
        ko, vo = ops.radix_sort_pairs(ctx, keys, vals, 0, bits)
        ref_k, ref_i = torch.sort(keys.to(torch.int64) & ((1 << bits) - 1), stable=True)
        ok = torch.equal(ko.to(torch.int64) & 0xFFFFFFFF, ref_k) and torch.equal(vo.to(torch.int64), ref_i)
        if not ok:
            bad += 1
            print("MISMATCH rep", rep, "n", n, "bits", bits, flush=True)
print("reps", reps, "mismatches", bad)
