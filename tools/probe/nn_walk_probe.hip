// The matching kernel's tile walk, rebuilt piece by piece: which piece costs the MFMA pipe its time?
//   base  : 24 MFMAs per tile in two chains of twelve (fresh accumulators per tile), one score read per chain
//   +EPI  : the v_max3 tree + compare + two selects per chain
//   +LOAD : the lane's 48 bytes of the next tile from a 18.9 MB array, one tile ahead (register double buffer)
// 4080 waves (1020 workgroups of 4), 73 tiles each: the shape of the 3072 x 196608 query.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/probe/nn_walk_probe.hip -o tools/probe/nn_walk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ float max16(const f32x16& c) {
    const float m0 = max3f(c[0], c[1], c[2]), m1 = max3f(c[3], c[4], c[5]), m2 = max3f(c[6], c[7], c[8]);
    const float m3 = max3f(c[9], c[10], c[11]), m4 = max3f(c[12], c[13], c[14]);
    const float m5 = max3f(m0, m1, m2), m6 = max3f(m3, m4, c[15]);
    return m5 > m6 ? m5 : m6;
}
template <int EPI, int LOAD, int UNR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ DB, const float* __restrict__ Q, int tiles, int S, float* out) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int group = wid / S, seg = wid - group * S;
    float q[2][12];
    for (int t = 0; t < 2; ++t)
        for (int k2 = 0; k2 < 12; ++k2) q[t][k2] = Q[((group * 64 + t * 32 + j) * 24 + 12 * h + k2) & 0xFFFFF];
    float best[2] = {-1e30f, -1e30f}; int bt[2] = {-1, -1};
    const int tile0 = seg * tiles;
    const float* dbl = DB + (int64_t)j * 24 + 12 * h;
    float4 nx, ny, nz;
    { const float4* src = (const float4*)(dbl + (int64_t)tile0 * 768); nx = src[0]; ny = src[1]; nz = src[2]; }
#pragma unroll UNR
    for (int tile = tile0; tile < tile0 + tiles; ++tile) {
        const float4 x = nx, y = ny, z = nz;
        if (LOAD && tile + 1 < tile0 + tiles) { const float4* src = (const float4*)(dbl + (int64_t)(tile + 1) * 768); nx = src[0]; ny = src[1]; nz = src[2]; }
        const float a[12] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y, z.z, z.w};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], q[t][k2], c, 0, 0, 0);
            const float tm = EPI ? max16(c) : c[0];
            const bool better = tm > best[t];
            best[t] = better ? tm : best[t]; bt[t] = better ? tile : bt[t];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = best[0] + best[1] + bt[0] + bt[1];
}
template <int EPI, int LOAD, int UNR>
void run(const float* DB, const float* Q, float* out, const char* name, int tiles = 73, int blocks = 1020) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k<EPI, LOAD, UNR><<<blocks, 256>>>(DB, Q, tiles, 6400 / tiles, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 4.0 * blocks * tiles * 24 * 4096.0;
    printf("%-28s %.1f us  %.1f TFLOP/s\n", name, best * 1e3, flop / best / 1e9);
}
int main() {
    float *DB, *Q, *out;
    const size_t nDB = (size_t)6400 * 768;
    hipMalloc(&DB, nDB * 4); hipMalloc(&Q, (1 << 20) * 4); hipMalloc(&out, 1020 * 256 * 4);
    float* hbuf = (float*)malloc(nDB * 4);
    unsigned x = 777u;
    for (size_t i = 0; i < nDB; ++i) { x = x * 1664525u + 1013904223u; hbuf[i] = (x >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    hipMemcpy(DB, hbuf, nDB * 4, hipMemcpyHostToDevice); hipMemcpy(Q, hbuf, (1 << 20) * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 1, 1>(DB, Q, out, "full, 1024 blocks", 73, 1024);
        run<1, 1, 1>(DB, Q, out, "full, 1020 blocks", 73, 1020);
        run<1, 1, 1>(DB, Q, out, "full, 1016 blocks", 73, 1016);
        run<1, 1, 1>(DB, Q, out, "full, 1008 blocks", 73, 1008);
    }
    run<0, 0, 1>(DB, Q, out, "base");
    run<1, 0, 1>(DB, Q, out, "base + epilogue");
    run<0, 1, 1>(DB, Q, out, "base + loads");
    run<1, 1, 1>(DB, Q, out, "base + epilogue + loads");
    run<1, 1, 2>(DB, Q, out, "  ... unrolled x2");
    run<0, 0, 2>(DB, Q, out, "base unrolled x2");
    run<0, 0, 1>(DB, Q, out, "base, 292 tiles per wave", 292);
    run<0, 0, 1>(DB, Q, out, "base, 1024 blocks", 73, 1024);
    run<0, 0, 1>(DB, Q, out, "base, 512 blocks (2/SIMD)", 73, 512);
    run<0, 0, 1>(DB, Q, out, "base, 256 blocks (1/SIMD)", 73, 256);
    run<0, 0, 1>(DB, Q, out, "base, 2048 blocks (2 rounds)", 73, 2048);
    return 0;
}
