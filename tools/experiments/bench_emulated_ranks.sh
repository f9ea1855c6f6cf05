#!/bin/bash
# Crash test of bench.py's N > 1 path on a ONE-GPU box: N processes share cuda:0, torch.distributed on gloo, the library's
# communicator over tests/fake_rccl.  The printed value is NOT a measurement.  usage: bench_emulated_ranks.sh [N=2]
N=${1:-2}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export ST3R_BENCH_EMULATE_RANKS=1 ST3R_RCCL_LIB=$PWD/tests/_build/libfake_rccl.so
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus $N --steps 6 --warmup 2 > gpurun_out/bench_emulated_n$N.json 2> gpurun_out/bench_emulated_n$N.err
echo "rc=$?"; wc -l gpurun_out/bench_emulated_n$N.json; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_emulated_n$N.json')); print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(d['per_rank']); print(d['config']['parallelism'])"
tail -5 gpurun_out/bench_emulated_n$N.err
