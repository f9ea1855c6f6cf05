// What does v_mfma_f32_32x32x2_f32 sustain on this box, as a function of independent accumulator chains per wave and of
// resident waves per SIMD?  (The matching kernel walks 24 MFMAs per tile in two chains of twelve, four waves per SIMD.)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/probe/mfma_f32_peak.hip -o tools/probe/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// DATA: the operands are 12 + 12 random numbers per lane (unit-normal-ish) instead of two near-constants: the multipliers'
// switching activity -- and with it the power the chip draws and the clock it sustains -- depends on the data.
template <int CH, int RESTART>
__global__ __launch_bounds__(256) void kd(int iters, const float* __restrict__ rnd, float* out) {
    float a[12], b[CH][12];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    for (int k2 = 0; k2 < 12; ++k2) {
        a[k2] = rnd[(gid * 12 + k2) & 0xFFFFF];
        for (int i = 0; i < CH; ++i) b[i][k2] = rnd[(gid * 12 + k2 + 4099 * (i + 1)) & 0xFFFFF];
    }
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (RESTART) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { keep += c[i][0]; c[i] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
        }
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2)
#pragma unroll
            for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], b[i][k2], c[i], 0, 0, 0);
    }
    for (int i = 0; i < CH; ++i) keep += c[i][1];
    out[gid] = keep;
}
template <int CH, int RESTART>   // CH chains; RESTART: every 12 MFMAs a chain starts again from zero (like a new tile)
__global__ __launch_bounds__(256) void k(int iters, float* out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (RESTART) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { keep += c[i][0]; c[i] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
        }
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2)
#pragma unroll
            for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + k2, b + i, c[i], 0, 0, 0);
    }
    for (int i = 0; i < CH; ++i) keep += c[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = keep;
}
template <int CH, int RESTART>
void run_data(int waves_per_simd, const float* rnd, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 8000 / CH;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kd<CH, RESTART><<<256 * waves_per_simd, 256>>>(iters, rnd, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * waves_per_simd * 4 * (double)iters * 12 * CH * 4096.0;
    printf("RANDOM operands, chains %d restart %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", CH, RESTART, waves_per_simd, ms, flop / ms / 1e9);
}
template <int CH, int RESTART>
void run(int waves_per_simd, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000 / CH;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<CH, RESTART><<<256 * waves_per_simd, 256>>>(iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * waves_per_simd * 4 * (double)iters * 12 * CH * 4096.0;
    printf("chains %d restart %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", CH, RESTART, waves_per_simd, ms, flop / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    float* rnd; hipMalloc(&rnd, (1 << 20) * 4);
    {
        float* h = (float*)malloc((1 << 20) * 4);
        unsigned x = 12345u;
        for (int i = 0; i < (1 << 20); ++i) {   // sum of four uniforms, centred: roughly normal, sigma 0.58
            float sacc = 0;
            for (int j = 0; j < 4; ++j) { x = x * 1664525u + 1013904223u; sacc += (x >> 8) * (1.0f / 16777216.0f); }
            h[i] = sacc - 2.0f;
        }
        hipMemcpy(rnd, h, (1 << 20) * 4, hipMemcpyHostToDevice); free(h);
    }
    for (int w : {4, 8}) { run_data<2, 1>(w, rnd, out); }
    run_data<4, 1>(4, rnd, out);
    for (int w : {1, 2, 4, 8}) { run<1, 0>(w, out); run<2, 0>(w, out); run<2, 1>(w, out); if (w <= 4) run<4, 1>(w, out); }
    return 0;
}
