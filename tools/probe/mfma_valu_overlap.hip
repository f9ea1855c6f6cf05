// Do MFMA instructions of one wave overlap with VALU instructions of OTHER waves on the same SIMD?
// Launch 8 waves per SIMD (256-thread blocks x 8 per CU): mode 0 = all waves VALU fma chains, mode 1 = all waves MFMA,
// mode 2 = half the waves (odd wave ids) MFMA and half VALU.  If the pipes overlap, mode 2 takes ~max(half0, half1);
// if MFMA issue occupies the VALU, mode 2 takes ~half0 + half1.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o tools/probe/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>   // 0: fp32 16x16x4, 1: bf16 16x16x32
__global__ __launch_bounds__(256) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 1 || (mode == 2 && (wave & 1));
    const bool do_valu = mode == 0 || (mode == 2 && !(wave & 1));
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf16x8 va = {1, 2, 3, 4, 5, 6, 7, 8}, vb = {8, 7, 6, 5, 4, 3, 2, 1};
    if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {   // 64 independent-ish fmas per iteration (4 chains)
                c0 = __builtin_fmaf(a, b, c0); c1 = __builtin_fmaf(a, b, c1); c2 = __builtin_fmaf(a, b, c2); c3 = __builtin_fmaf(a, b, c3);
            }
        }
    }
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // 8 MFMAs per iteration (2 chains)
                if (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, va, acc1, 0, 0, 0);
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + acc0[0] + acc1[1];
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) k<0><<<256 * 8, 256>>>(mode, iters, out); else k<1><<<256 * 8, 256>>>(mode, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%s mode %d (%s): %.3f ms\n", kind ? "bf16 16x16x32" : "fp32 16x16x4 ", mode,
                   mode == 0 ? "all waves VALU" : mode == 1 ? "all waves MFMA" : "half VALU, half MFMA", ms);
        }
    return 0;
}
