// C7: the refinement hooks of gsplat.MCMCStrategy (defaults) that starster/gs.py:43-45,146-147,163-164
// drives -- relocate dead Gaussians, grow by 5 %, perturb the means -- as device kernels with a
// counter-based generator, so that view-sharded replicas (one process per GPU) make bit-identical
// decisions without exchanging anything.
//
// What is kept from the reference: a Gaussian is dead when sigmoid(opacity) <=
// min_opacity; sources are drawn with replacement with probability proportional to sigmoid(opacity) (alive
// ones only when relocating); a source drawn r-1 times gets ratio r = min(r, 51) and
//     o' = 1 - (1 - o)^(1/r),   s' = s * o / sum_{i=1..r} sum_{k<i} C(i-1,k) (-1)^k o'^(k+1) / sqrt(k+1),
// stored back as logit(clamp(o', min_opacity, 1 - eps)) and log(s'); the dead (or appended) rows then copy
// every parameter of their source; the Adam moments of the SOURCES are zeroed when relocating and left
// alone when growing (appended rows start from zero).  Opacities are read as logits and scales as logs
// here although the renderer uses the same tensors raw (App. B-1) -- reproduced on purpose.
//
// What differs: torch.multinomial / torch.randn streams cannot be reproduced; draws come from Philox4x32-10
// keyed by (seed) with counter (index, stream, step).  Sampling is done in exact integer arithmetic
// (24-bit fixed-point weights, 64-bit prefix sums, mulhi of a 64-bit draw) so the CPU oracle can replay it
// bit for bit.
#include "common.h"

#include "scan.h"

#define MCMC_NMAX 51
#define STREAM_RELOCATE 0u
#define STREAM_ADD 1u
#define STREAM_NOISE 2u

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                     uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// weights (24-bit fixed point of sigmoid(opacity)) and the dead mask
__global__ __launch_bounds__(256) void k_mcmc_weights(int N, const float* __restrict__ opacities, float min_opacity,
                                                      int relocating, uint64_t* __restrict__ w,
                                                      uint32_t* __restrict__ dead) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const float p = sigmoidf_(opacities[g]);
    const bool is_dead = relocating && (p <= min_opacity);
    dead[g] = is_dead ? 1u : 0u;
    w[g] = is_dead ? 0ull : (uint64_t)(p * 16777216.0f);
}

__device__ __forceinline__ int draw_index(const uint64_t* __restrict__ cum, int N, uint64_t i, uint32_t stream_id,
                                          uint32_t step, uint64_t seed) {
    const U4 r = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), stream_id, step, (uint32_t)seed,
                               (uint32_t)(seed >> 32));
    const uint64_t r64 = ((uint64_t)r.x << 32) | r.y;
    const uint64_t t = __umul64hi(r64, cum[N - 1]);  // uniform in [0, total)
    int lo = 0, hi = N;                              // first g with cum[g] > t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo < N ? lo : N - 1;
}

// relocate: the i-th dead Gaussian (ascending index) takes draw i
__global__ __launch_bounds__(256) void k_mcmc_draw_dead(int N, const uint64_t* __restrict__ cum,
                                                        const uint32_t* __restrict__ dead,
                                                        const uint32_t* __restrict__ rank, uint32_t step, uint64_t seed,
                                                        int32_t* __restrict__ sampled, int32_t* __restrict__ dead_ids,
                                                        uint32_t* __restrict__ count) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N || !dead[g]) return;
    if (cum[N - 1] == 0) { sampled[rank[g]] = -1; dead_ids[rank[g]] = g; return; }  // nothing alive
    const int src = draw_index(cum, N, rank[g], STREAM_RELOCATE, step, seed);
    sampled[rank[g]] = src;
    dead_ids[rank[g]] = g;
    atomicAdd(&count[src], 1u);
}

// grow: draw i fills appended row N + i
__global__ __launch_bounds__(256) void k_mcmc_draw_new(int N, int n_new, const uint64_t* __restrict__ cum,
                                                       uint32_t step, uint64_t seed, int32_t* __restrict__ sampled,
                                                       uint32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_new) return;
    const int src = draw_index(cum, N, i, STREAM_ADD, step, seed);
    sampled[i] = src;
    atomicAdd(&count[src], 1u);
}

// new opacity / scale of every drawn source (gsplat compute_relocation + clamp + logit/log)
__global__ __launch_bounds__(256) void k_mcmc_update_sources(int N, const uint32_t* __restrict__ count,
                                                             const float* __restrict__ binoms,
                                                             float* __restrict__ opacities, float* __restrict__ scales,
                                                             float min_opacity, float* __restrict__ adam_m,
                                                             float* __restrict__ adam_v) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const uint32_t c = count[g];
    if (c == 0) return;
    const int n_idx = min((int)c + 1, MCMC_NMAX);
    const float o = sigmoidf_(opacities[g]);
    float new_o = 1.0f - powf(1.0f - o, 1.0f / (float)n_idx);
    float denom = 0.0f;
    for (int i = 1; i <= n_idx; ++i) {
        for (int k = 0; k < i; ++k) {
            const float sgn = (k & 1) ? -1.0f : 1.0f;
            const float term = (sgn / sqrtf((float)(k + 1))) * powf(new_o, (float)(k + 1));
            denom += binoms[(i - 1) * MCMC_NMAX + k] * term;
        }
    }
    const float coeff = o / denom;
#pragma unroll
    for (int j = 0; j < 3; ++j) scales[3 * g + j] = logf(coeff * expf(scales[3 * g + j]));
    new_o = fminf(fmaxf(new_o, min_opacity), 1.0f - 1.1920929e-07f);
    opacities[g] = logf(new_o / (1.0f - new_o));
    if (adam_m) {  // [23N] blocks: means 3 | quats 4 | scales 3 | opacities 1 | sh rows 0..3 12
        const int64_t Nl = N;
        const int64_t base[5] = {0, 3 * Nl, 7 * Nl, 10 * Nl, 11 * Nl};
        const int width[5] = {3, 4, 3, 1, 12};
#pragma unroll
        for (int b = 0; b < 5; ++b)
            for (int j = 0; j < width[b]; ++j) {
                adam_m[base[b] + (int64_t)g * width[b] + j] = 0.f;
                adam_v[base[b] + (int64_t)g * width[b] + j] = 0.f;
            }
    }
}

// dst rows copy every parameter of their (already updated) source; 8 lanes per row
__global__ __launch_bounds__(256) void k_mcmc_copy_rows(int n_rows, const int32_t* __restrict__ n_rows_dev,
                                                        const int32_t* __restrict__ sampled,
                                                        const int32_t* __restrict__ dst_ids, int dst_base,
                                                        float* __restrict__ means, float* __restrict__ quats,
                                                        float* __restrict__ scales, float* __restrict__ opacities,
                                                        float* __restrict__ sh0, float* __restrict__ shN,
                                                        int shN_floats) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    const int n = n_rows_dev ? *n_rows_dev : n_rows;
    if (i >= n) return;
    const int src = sampled[i];
    if (src < 0) return;
    const int64_t dst = dst_ids ? dst_ids[i] : (int64_t)dst_base + i;
    if (l < 3) { means[3 * dst + l] = means[3 * (int64_t)src + l]; scales[3 * dst + l] = scales[3 * (int64_t)src + l]; }
    if (l < 4) quats[4 * dst + l] = quats[4 * (int64_t)src + l];
    if (l == 4) opacities[dst] = opacities[src];
    if (sh0 && l >= 5) sh0[3 * dst + (l - 5)] = sh0[3 * (int64_t)src + (l - 5)];
    for (int j = l; j < shN_floats; j += 8) shN[dst * shN_floats + j] = shN[(int64_t)src * shN_floats + j];
}

__global__ void k_mcmc_last_count(int N, const uint32_t* __restrict__ dead, const uint32_t* __restrict__ rank,
                                  int32_t* __restrict__ n_dead) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_dead = (int32_t)(rank[N - 1] + dead[N - 1]);
}

// inject_noise_to_position: means += Sigma(quats, exp(scales)) @ (randn * gate(opacity) * scaler)
__global__ __launch_bounds__(256) void k_mcmc_noise(int N, float* __restrict__ means, const float* __restrict__ quats,
                                                    const float* __restrict__ scales,
                                                    const float* __restrict__ opacities, float scaler, uint32_t step,
                                                    uint64_t seed, uint32_t row_offset) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    // the draw belongs to the Gaussian's global row: a shard of the rows perturbs exactly like the whole set
    const U4 r = philox4x32_10((uint32_t)g + row_offset, 0u, STREAM_NOISE, step, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float inv24 = 1.0f / 16777216.0f;
    const float u1 = ((float)(r.x >> 8) + 0.5f) * inv24, u2 = ((float)(r.y >> 8) + 0.5f) * inv24;
    const float u3 = ((float)(r.z >> 8) + 0.5f) * inv24, u4 = ((float)(r.w >> 8) + 0.5f) * inv24;
    const float TWO_PI = 6.283185307179586f;
    const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float nz[3] = {ra * cosf(TWO_PI * u2), ra * sinf(TWO_PI * u2), rb * cosf(TWO_PI * u4)};
    const float op = sigmoidf_(opacities[g]);
    const float gate = 1.0f / (1.0f + expf(-100.0f * ((1.0f - op) - 0.995f)));
    const float k = gate * scaler;
    float qw = quats[4 * g], qx = quats[4 * g + 1], qy = quats[4 * g + 2], qz = quats[4 * g + 3];
    const float inv = 1.0f / sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    qw *= inv; qx *= inv; qy *= inv; qz *= inv;
    const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                        2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                        2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
    const float s[3] = {expf(scales[3 * g]), expf(scales[3 * g + 1]), expf(scales[3 * g + 2])};
    // Sigma v = R diag(s^2) R^T v
    float t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = (R[j] * nz[0] + R[3 + j] * nz[1] + R[6 + j] * nz[2]) * k * s[j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) means[3 * g + i] += R[3 * i] * t[0] + R[3 * i + 1] * t[1] + R[3 * i + 2] * t[2];
}

struct McmcScratch { uint64_t* cum; uint32_t *dead, *rank, *count; int32_t *sampled, *dead_ids, *n_dead; float* binoms; };

static int mcmc_scratch(st3r_ctx* ctx, hipStream_t s, int64_t N, int64_t n_rows, McmcScratch* o) {
    void* p; int rc;
    if ((rc = st3r_arena_get(ctx, SLOT_MCMC_CUM, sizeof(uint64_t) * (size_t)N, &p))) return rc;
    o->cum = (uint64_t*)p;
    if ((rc = st3r_arena_get(ctx, SLOT_MCMC_DEAD, sizeof(uint32_t) * (size_t)N, &p))) return rc;
    o->dead = (uint32_t*)p;
    if ((rc = st3r_arena_get(ctx, SLOT_MCMC_RANK, sizeof(uint32_t) * (size_t)N, &p))) return rc;
    o->rank = (uint32_t*)p;
    if ((rc = st3r_arena_get(ctx, SLOT_MCMC_COUNT, sizeof(uint32_t) * (size_t)N, &p))) return rc;
    o->count = (uint32_t*)p;
    if ((rc = st3r_arena_get(ctx, SLOT_MCMC_SAMPLED, sizeof(int32_t) * (size_t)(2 * n_rows + 4), &p))) return rc;
    o->sampled = (int32_t*)p; o->dead_ids = o->sampled + n_rows; o->n_dead = o->sampled + 2 * n_rows;
    int grown = 0;
    if ((rc = st3r_arena_get2(ctx, SLOT_MCMC_BINOMS, sizeof(float) * MCMC_NMAX * MCMC_NMAX, &p, &grown))) return rc;
    o->binoms = (float*)p;
    if (grown) {  // Pascal triangle in doubles (exact up to C(50,25) ~ 1.3e14), rounded once to float like the
                  // float32 table gsplat builds from math.comb
        static float host_tab[MCMC_NMAX * MCMC_NMAX];
        static double tri[MCMC_NMAX][MCMC_NMAX];
        for (int n = 0; n < MCMC_NMAX; ++n)
            for (int k = 0; k < MCMC_NMAX; ++k) {
                tri[n][k] = (k == 0) ? 1.0 : (k > n ? 0.0 : tri[n - 1][k - 1] + (k <= n - 1 ? tri[n - 1][k] : 0.0));
                host_tab[n * MCMC_NMAX + k] = (float)tri[n][k];
            }
        HIP_TRY(hipMemcpyAsync(o->binoms, host_tab, sizeof(host_tab), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    HIP_TRY(hipMemsetAsync(o->count, 0, sizeof(uint32_t) * (size_t)N, s));
    return ST3R_OK;
}

static int scan_weights(st3r_ctx* ctx, hipStream_t s, int64_t N, McmcScratch& sc, bool with_rank) {
    // inclusive 64-bit prefix sums of the fixed-point weights (in place) and, for the relocation, the exclusive
    // ranks of the dead Gaussians: hand-written three-phase scans (scan.h)
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SCAN_TMP, st3r_scan::scratch_bytes<uint64_t>(N), &tmp);
    if (rc) return rc;
    st3r_scan::scan<uint64_t, false>(s, sc.cum, sc.cum, N, tmp);
    if (with_rank) st3r_scan::scan<uint32_t, true>(s, sc.dead, sc.rank, N, tmp);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_mcmc_relocate(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                                   float* opacities, float* sh0, float* shN, int shN_floats, float* adam_m,
                                   float* adam_v, float min_opacity, uint64_t seed, uint32_t step,
                                   int64_t* n_dead_host) {
    ARG_CHECK(ctx && N > 0 && means && quats && scales && opacities && shN && shN_floats >= 12);
    ARG_CHECK((adam_m == nullptr) == (adam_v == nullptr));
    hipStream_t s = (hipStream_t)stream;
    McmcScratch sc;
    int rc = mcmc_scratch(ctx, s, N, N, &sc);
    if (rc) return rc;
    const int blocks = ceil_div(N, 256);
    hipLaunchKernelGGL(k_mcmc_weights, dim3(blocks), dim3(256), 0, s, N, opacities, min_opacity, 1, sc.cum, sc.dead);
    LAUNCH_CHECK();
    if ((rc = scan_weights(ctx, s, N, sc, true))) return rc;
    hipLaunchKernelGGL(k_mcmc_last_count, dim3(1), dim3(64), 0, s, N, sc.dead, sc.rank, sc.n_dead);
    hipLaunchKernelGGL(k_mcmc_draw_dead, dim3(blocks), dim3(256), 0, s, N, sc.cum, sc.dead, sc.rank, step, seed,
                       sc.sampled, sc.dead_ids, sc.count);
    hipLaunchKernelGGL(k_mcmc_update_sources, dim3(blocks), dim3(256), 0, s, N, sc.count, sc.binoms, opacities, scales,
                       min_opacity, adam_m, adam_v);
    // the row count stays on the device: every potential row gets 8 lanes and the kernel reads n_dead itself
    hipLaunchKernelGGL(k_mcmc_copy_rows, dim3(ceil_div((int64_t)N * 8, 256)), dim3(256), 0, s, N, sc.n_dead, sc.sampled,
                       sc.dead_ids, 0, means, quats, scales, opacities, sh0, shN, shN_floats);
    LAUNCH_CHECK();
    if (n_dead_host) {
        HIP_TRY(hipMemcpyAsync(ctx->pinned, sc.n_dead, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *n_dead_host = (int64_t)((int32_t*)ctx->pinned)[0];
    }
    return ST3R_OK;
}

ST3R_EXPORT int st3r_mcmc_add(st3r_ctx* ctx, void* stream, int N, int n_new, float* means, float* quats, float* scales,
                              float* opacities, float* sh0, float* shN, int shN_floats, float min_opacity,
                              uint64_t seed, uint32_t step) {
    ARG_CHECK(ctx && N > 0 && n_new >= 0 && means && quats && scales && opacities && shN && shN_floats >= 12);
    if (n_new == 0) return ST3R_OK;
    hipStream_t s = (hipStream_t)stream;
    McmcScratch sc;
    int rc = mcmc_scratch(ctx, s, N, n_new, &sc);
    if (rc) return rc;
    const int blocks = ceil_div(N, 256);
    hipLaunchKernelGGL(k_mcmc_weights, dim3(blocks), dim3(256), 0, s, N, opacities, min_opacity, 0, sc.cum, sc.dead);
    LAUNCH_CHECK();
    if ((rc = scan_weights(ctx, s, N, sc, false))) return rc;
    hipLaunchKernelGGL(k_mcmc_draw_new, dim3(ceil_div(n_new, 256)), dim3(256), 0, s, N, n_new, sc.cum, step, seed,
                       sc.sampled, sc.count);
    hipLaunchKernelGGL(k_mcmc_update_sources, dim3(blocks), dim3(256), 0, s, N, sc.count, sc.binoms, opacities, scales,
                       min_opacity, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_mcmc_copy_rows, dim3(ceil_div((int64_t)n_new * 8, 256)), dim3(256), 0, s, n_new,
                       (const int32_t*)nullptr, sc.sampled, (const int32_t*)nullptr, N, means, quats, scales, opacities,
                       sh0, shN, shN_floats);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_mcmc_noise_rows(st3r_ctx* ctx, void* stream, int n, int64_t row_offset, float* means,
                                     const float* quats, const float* scales, const float* opacities, float scaler,
                                     uint64_t seed, uint32_t step) {
    ARG_CHECK(ctx && n >= 0 && row_offset >= 0 && row_offset + n <= 0xFFFFFFFFLL && means && quats && scales && opacities);
    if (n == 0) return ST3R_OK;
    hipLaunchKernelGGL(k_mcmc_noise, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n, means, quats, scales,
                       opacities, scaler, step, seed, (uint32_t)row_offset);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_mcmc_noise(st3r_ctx* ctx, void* stream, int N, float* means, const float* quats,
                                const float* scales, const float* opacities, float scaler, uint64_t seed,
                                uint32_t step) {
    return st3r_mcmc_noise_rows(ctx, stream, N, 0, means, quats, scales, opacities, scaler, seed, step);
}
