// Replay of the blend-forward trip body (k_blend_fwd<true>, gs_blend.hip) without global memory: what does one
// (record, wave) trip cost when nothing but the loop itself runs, and which ingredient costs what?
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops tools/probe/trip_replay.hip -o tools/probe/trip_replay
// 2048 workgroups x 256 threads = 8 waves per SIMD on 256 CUs, 256 synthetic records in LDS, every wave walks
// ROUNDS x (trips of the mode).  Reported: ns per trip per SIMD (kernel time x 1024 SIMDs / wave trips) and the
// wave's own shader cycles per trip (s_memtime).
// Modes
//   0  the production loop: 64-bit relevance word walked as two halves with scalar bit scans, 2 x ds_read_b128 +
//      ds_read_b32 at a wave-uniform address, compares into SGPR masks, select, rare saturation branch
//   1  mode 0 without the bit scans: t = 0, 1, 2, ... (scalar counter)
//   2  mode 1 without the LDS reads: the record sits in registers (made opaque per trip)
//   3  mode 0 with v_cmpx / exec masking instead of mask + select (no v_cndmask, no s_nop)
//   4  mode 1 with only the arithmetic: no tests at all (al = al0), lower bound of the body
//   5  cells: each 16-lane row walks its own list (ds_read_u16 of a byte offset, then the three record reads at
//      row-uniform addresses), trip count = the longest of the four lists
//   6  per-lane lists: every lane its own record (per-lane gather)
//   7  mode 0, but the upper half of the wave is switched off in exec (does a half-empty wave issue faster?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>

#define NREC 256
#define ROUNDS 24

__device__ __forceinline__ uint64_t mask_not_less(float a, float b) {
    uint64_t m; asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m;
}
__device__ __forceinline__ uint64_t mask_not_positive(float a) {
    uint64_t m; asm("v_cmp_nlt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a)); return m;
}
__device__ __forceinline__ float zero_unless(uint64_t m, float x) {
    float r; asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m)); return r;
}

struct Px { float px, py, thr, T, r, g, b; int cur; };

// the production trip body (TRAIN variant), given the three LDS words of the record
__device__ __forceinline__ uint64_t trip(const float4 a, const float4 q, const float cb, Px& s, int gidx) {
    const float dx = a.x - s.px, dy = a.y - s.py;
    const float P = dx * (a.w * dx + q.x * dy) + q.y * dy * dy;
    const float al0 = fminf(0.999f, a.z * __builtin_amdgcn_exp2f(P));
    const uint64_t okm = mask_not_positive(P) & mask_not_less(al0, s.thr);
    const float al = zero_unless(okm, al0);
    const float nT = s.T * (1.0f - al);
    const bool stop = nT <= 1e-4f;
    float vis = al * s.T;
    uint64_t tm = okm;
    float Tn = nT;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(stop) != 0, 0)) {
        s.thr = stop ? __builtin_inff() : s.thr;
        vis = stop ? 0.0f : vis;
        Tn = stop ? s.T : nT;
        s.cur = stop ? gidx : s.cur;
        tm = okm & ~__builtin_amdgcn_ballot_w64(stop);
    }
    s.T = Tn;
    s.r += q.z * vis; s.g += q.w * vis; s.b += cb * vis;
    return tm;
}

// exec-masked form: lanes that fail the tests skip the updates instead of blending alpha = 0
__device__ __forceinline__ void trip_exec(const float4 a, const float4 q, const float cb, Px& s, int gidx) {
    const float dx = a.x - s.px, dy = a.y - s.py;
    const float P = dx * (a.w * dx + q.x * dy) + q.y * dy * dy;
    const float al = fminf(0.999f, a.z * __builtin_amdgcn_exp2f(P));
    if (!(P > 0.f) && !(al < s.thr)) {
        const float nT = s.T * (1.0f - al);
        if (nT <= 1e-4f) { s.thr = __builtin_inff(); s.cur = gidx; }
        else { const float vis = al * s.T; s.T = nT; s.r += q.z * vis; s.g += q.w * vis; s.b += cb * vis; }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_replay(const float4* __restrict__ recs, float* __restrict__ out,
                                                unsigned long long* __restrict__ cyc, const uint64_t* __restrict__ masks) {
    __shared__ float4 sR[(NREC + 1) * 3];
    __shared__ uint16_t sList[16][NREC + 8];
    __shared__ uint64_t sMask[4][4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < (NREC + 1) * 3; i += 256) sR[i] = recs[i];
    if (threadIdx.x < 16) sMask[threadIdx.x >> 2][threadIdx.x & 3] = masks[threadIdx.x];
    // lists: cell c takes every record whose index has (t % 3 != c % 3) -> ~2/3 of the records, padded with the sentinel
    for (int c = threadIdx.x; c < 16; c += 256) {
        int n = 0;
        for (int t = 0; t < NREC; ++t) if ((t + c) % 3 != 0) sList[c][n++] = (uint16_t)(t * 48);
        for (; n < NREC + 8; ++n) sList[c][n] = (uint16_t)(NREC * 48);
    }
    __syncthreads();
    Px s;
    s.px = (float)(((w & 1) << 3) + (lane & 7)) + 0.5f; s.py = (float)(((w >> 1) << 3) + (lane >> 3)) + 0.5f;
    s.thr = 1.f / 255.f; s.T = 1.f; s.r = s.g = s.b = 0.f; s.cur = 0x7fffffff;
    uint64_t cont = 0;
    long long trips = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int round = 0; round < ROUNDS; ++round) {
        s.T = 1.0f; s.thr = 1.f / 255.f;   // keeps the pixels live (the rare branch stays rare)
        if (MODE == 0 || MODE == 3 || MODE == 7) {
            if (MODE == 7) asm volatile("s_mov_b64 exec, 0xffffffff");
#pragma unroll 1
            for (int jj = 0; jj < 4; ++jj) {
                uint64_t m64 = sMask[w][jj];
                m64 = ((uint64_t)__builtin_amdgcn_readfirstlane((unsigned)(m64 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)m64);
#pragma unroll 1
                for (int hh = 0; hh < 2; ++hh) {
                    uint32_t m = hh ? (uint32_t)(m64 >> 32) : (uint32_t)m64;
                    uint32_t c32 = 0;
                    while (m) {
                        const int bit = __builtin_ctz(m);
                        m &= m - 1;
                        const int t = jj * 64 + hh * 32 + bit;
                        const float4 a = sR[3 * t], q = sR[3 * t + 1];
                        const float cb = sR[3 * t + 2].x;
                        if (MODE == 3) trip_exec(a, q, cb, s, t);
                        else c32 |= trip(a, q, cb, s, t) ? (1u << bit) : 0u;
                        ++trips;
                    }
                    cont |= (uint64_t)c32 << (32 * hh);
                }
            }
            if (MODE == 7) asm volatile("s_mov_b64 exec, -1");
        } else if (MODE == 1 || MODE == 2 || MODE == 4) {
            float4 a = sR[0], q = sR[1]; float cb = sR[2].x;
#pragma unroll 1
            for (int t = 0; t < NREC; ++t) {
                if (MODE == 2) {
                    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
                    asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "+v"(cb));
                } else { a = sR[3 * t]; q = sR[3 * t + 1]; cb = sR[3 * t + 2].x; }
                if (MODE == 4) {
                    const float dx = a.x - s.px, dy = a.y - s.py;
                    const float P = dx * (a.w * dx + q.x * dy) + q.y * dy * dy;
                    const float al = a.z * __builtin_amdgcn_exp2f(P);
                    const float vis = al * s.T;
                    s.T = s.T * (1.0f - al);
                    s.r += q.z * vis; s.g += q.w * vis; s.b += cb * vis;
                } else cont |= trip(a, q, cb, s, t);
                ++trips;
            }
        } else if (MODE == 5) {
            const int c = 4 * w + (lane >> 4);
            const uint16_t* lp = sList[c];
            const int n = (NREC * 2) / 3 + 2;
            const char* base = reinterpret_cast<const char*>(sR);
#pragma unroll 4
            for (int k = 0; k < n; ++k) {
                const int off = lp[k];
                const float4 a = *reinterpret_cast<const float4*>(base + off);
                const float4 q = *reinterpret_cast<const float4*>(base + off + 16);
                const float cb = *reinterpret_cast<const float*>(base + off + 32);
                cont |= trip(a, q, cb, s, off);
                ++trips;
            }
        } else if (MODE == 6) {
            const char* base = reinterpret_cast<const char*>(sR);
            unsigned h = lane * 2654435761u;
#pragma unroll 1
            for (int k = 0; k < NREC / 2; ++k) {
                // neighbouring lanes mostly share a record (footprints are ~25 pixels): the record index depends on lane / 4
                h = h * 1664525u + 1013904223u;
                const int off = (((lane >> 2) * 5 + k * 3 + ((h >> 28) & 1)) & 255) * 48;
                const float4 a = *reinterpret_cast<const float4*>(base + off);
                const float4 q = *reinterpret_cast<const float4*>(base + off + 16);
                const float cb = *reinterpret_cast<const float*>(base + off + 32);
                cont |= trip(a, q, cb, s, off);
                ++trips;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s.r + s.g + s.b + s.T + (float)s.cur + (float)(cont & 1);
    if (lane == 0) { atomicAdd(&cyc[0], t1 - t0); atomicAdd(&cyc[1], (unsigned long long)trips); }
}

template <typename K, typename... A>
static float time_ms(K k, int blocks, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int MODE>
static void run(const char* name, const float4* recs, float* out, unsigned long long* cyc, const uint64_t* masks, int blocks) {
    hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k_replay<MODE>, dim3(blocks), dim3(256), 0, 0, recs, out, cyc, masks);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
    hipMemset(cyc, 0, 16);
    const float ms = time_ms(k_replay<MODE>, blocks, recs, out, cyc, masks);
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    // the timed launch ran twice (warm-up inside time_ms + timed): counters hold both
    const double trips_per_launch = (double)h[1] / 2.0;   // wave trips
    fprintf(stderr, "done %s\n", name);
    printf("%-44s %2d waves/SIMD  %7.3f ms  %6.2f ns/trip/SIMD  %7.1f wave-cycles/trip (x waves = %6.1f SIMD cycles)\n", name,
           blocks / 256, ms, ms * 1e6 * 1024.0 / trips_per_launch, (double)h[0] / (double)h[1],
           (double)h[0] / (double)h[1] / (blocks / 256.0) );
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // records: means scattered over a 16x16 tile (+- 4 px), sigma ~ 2.5 px, opacity 0.05 .. 1, colours
    float4 h[(NREC + 1) * 3];
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    const float L2E = 1.4426950408889634f;
    for (int t = 0; t < NREC; ++t) {
        const float sx = 1.5f + 2.0f * rnd(), sy = 1.5f + 2.0f * rnd(), rho = 0.6f * (rnd() - 0.5f);
        const float a = 1.f / (sx * sx * (1 - rho * rho)), c = 1.f / (sy * sy * (1 - rho * rho)), b = -rho / (sx * sy * (1 - rho * rho));
        h[3 * t] = make_float4(-4.f + 24.f * rnd(), -4.f + 24.f * rnd(), 0.05f + 0.45f * rnd(), -0.5f * L2E * a);
        h[3 * t + 1] = make_float4(-L2E * b, -0.5f * L2E * c, rnd(), rnd());
        h[3 * t + 2] = make_float4(rnd(), 0, 0, 0);
    }
    h[3 * NREC] = make_float4(0, 0, 0, 0); h[3 * NREC + 1] = make_float4(0, 0, 0, 0); h[3 * NREC + 2] = make_float4(0, 0, 0, 0);
    uint64_t hm[16];
    for (int i = 0; i < 16; ++i) { hm[i] = 0; for (int b = 0; b < 64; ++b) if ((b + i) % 3 != 0) hm[i] |= 1ull << b; }
    float4* recs; float* out; unsigned long long* cyc; uint64_t* masks;
    hipMalloc(&recs, sizeof(h)); hipMemcpy(recs, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&masks, sizeof(hm)); hipMemcpy(masks, hm, sizeof(hm), hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 256 * sizeof(float)); hipMalloc(&cyc, 16);
    for (int blocks : {2048, 1024, 256}) {
        run<0>("0 production loop", recs, out, cyc, masks, blocks);
        run<1>("1 no bit scan (sequential t)", recs, out, cyc, masks, blocks);
        run<2>("2 no bit scan, no LDS reads", recs, out, cyc, masks, blocks);
        run<3>("3 exec-masked tests (no select)", recs, out, cyc, masks, blocks);
        run<4>("4 arithmetic only (no tests)", recs, out, cyc, masks, blocks);
        run<5>("5 cells: per-row list + record", recs, out, cyc, masks, blocks);
        run<6>("6 per-lane gather", recs, out, cyc, masks, blocks);
        run<7>("7 production loop, upper half of exec off", recs, out, cyc, masks, blocks);
    }
    return 0;
}
