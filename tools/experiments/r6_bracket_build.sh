#!/bin/bash
# Round 6 bracket for the wave-per-8x8-tile formulation: what do the workgroup barriers of the two blend
# kernels cost?  Timing-only variants (results are WRONG: waves read records other waves have not staged yet) built from
# patched copies of the product sources; run with tools/experiments/abl.sh run (frozen scene).
set -u
rm -rf build_variants; mkdir -p build_variants
build() {  # $1 = index, $2 = label
  python -m starst3r_amd.build --force > /dev/null 2>&1 || echo "build failed: $2"
  cp starst3r_amd/libst3r_hip.so build_variants/v$1.so; echo "$2" > build_variants/v$1.txt
}
build 1 "product"
# backward: the three workgroup barriers of a round -> wave barriers
python - <<'PY'
import re
p='starst3r_amd/csrc/gs_blend.hip'; s=open(p).read()
a=s.index('void k_blend_bwd('); b=s.index('__global__ __launch_bounds__(256) void k_gather_vtile')
body=s[a:b]; assert body.count('__syncthreads();')==3
s=s[:a]+body.replace('__syncthreads();','__builtin_amdgcn_wave_barrier();')+s[b:]
open(p,'w').write(s)
PY
build 2 "bwd: no workgroup barriers"
git checkout starst3r_amd/csrc/gs_blend.hip
python - <<'PY'
p='starst3r_amd/csrc/gs_blend_cells.hip'; s=open(p).read()
a=s.index('void k_blend_fwd_cells('); body=s[a:]
assert body.count('__syncthreads();')==2 and body.count('__syncthreads_and(s.thr > 1.0f)')==1
body=body.replace('__syncthreads();','__builtin_amdgcn_wave_barrier();').replace('__syncthreads_and(s.thr > 1.0f)','(__builtin_amdgcn_ballot_w64(!(s.thr > 1.0f)) == 0)')
open(p,'w').write(s[:a]+body)
PY
build 3 "fwd: no workgroup barriers"
python - <<'PY'
p='starst3r_amd/csrc/gs_blend.hip'; s=open(p).read()
a=s.index('void k_blend_bwd('); b=s.index('__global__ __launch_bounds__(256) void k_gather_vtile')
s=s[:a]+s[a:b].replace('__syncthreads();','__builtin_amdgcn_wave_barrier();')+s[b:]
open(p,'w').write(s)
PY
build 4 "fwd + bwd: no workgroup barriers"
git checkout starst3r_amd/csrc/gs_blend.hip starst3r_amd/csrc/gs_blend_cells.hip
python -m starst3r_amd.build --force > /dev/null 2>&1
ls -la build_variants
