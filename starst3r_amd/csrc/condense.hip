// SURVEY 8(f) row 2 -- canonical-data condensation between the matching (path A) and the alignment (path B):
// what starster/reconstruct.py:101-106 gets from Mast3r's prepare_canonical_data / condense_data
// (mast3r/cloud_opt/sparse_ga.py [U]: the submodule is not vendored in the reference tree; the formulas below
// restate its published algorithm, oracle/condense_oracle.py is their CPU restatement).
//
//   k_canon_mean     per pixel: confidence-weighted mean of the n pointmaps an image got as "image 1" of a pair,
//                    weights w = conf - 0.999;  cconf = sum w^2 / sum w                    (canonical_view [U])
//   k_canon_angle    per pixel ('avg-angle' mode): inside its subsample x subsample block every prediction k gives
//                    an elevation angle of the pixel relative to the block centre, atan((z - zc) / |xy - xyc|);
//                    the angles are averaged with the weights, turned back into a depth at the mean radius and
//                    stored relative to the canonical depth of the block centre:  canon2 = 1 + depth / canon_z(c)
//   k_focal_weiszfeld  focal of a canonical pointmap given the principal point: closed-form L2 start, then 10
//                    re-weighted least-squares steps (weights 1 / |pixel - f xy/z|), clipped to
//                    [min_focal, max_focal] x the 60-degree-FOV focal      (dust3r estimate_focal_knowing_depth [U])
//   k_anchor_offsets per correspondence pixel: index of its block's core depth and the pixel's canon2 relative to the
//                    block centre's                                                         (anchor_depth_offsets [U])
// The problems are small (n <= C-1 maps of 512 x 384): everything is HBM/latency bound and deterministic (fixed-order
// sums); the focal solve is one workgroup.
#include "common.h"

__global__ __launch_bounds__(256) void k_canon_mean(int n, int HW, const float* __restrict__ ptmaps,
                                                    const float* __restrict__ confs, float* __restrict__ canon,
                                                    float* __restrict__ cconf) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f, sw2 = 0.f;
    for (int k = 0; k < n; ++k) {
        const float w = confs[(int64_t)k * HW + p] - 0.999f;
        const float* X = ptmaps + ((int64_t)k * HW + p) * 3;
        sx += w * X[0]; sy += w * X[1]; sz += w * X[2];
        sw += w; sw2 += w * w;
    }
    canon[3 * p + 0] = sx / sw; canon[3 * p + 1] = sy / sw; canon[3 * p + 2] = sz / sw;
    cconf[p] = sw2 / sw;
}

__global__ __launch_bounds__(256) void k_canon_angle(int n, int H, int W, int S, const float* __restrict__ ptmaps,
                                                     const float* __restrict__ confs,
                                                     const float* __restrict__ canon, float* __restrict__ canon2) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const int c = ((y / S) * S + S / 2) * W + (x / S) * S + S / 2;   // block centre
    float sa = 0.f, sw = 0.f, sr = 0.f;
    for (int k = 0; k < n; ++k) {
        const float* Xp = ptmaps + ((int64_t)k * HW + p) * 3;
        const float* Xc = ptmaps + ((int64_t)k * HW + c) * 3;
        const float zc = fmaxf(Xc[2], 1.1920928955078125e-07f);    // clip(min = finfo(float32).eps)
        const float dx = Xp[0] - Xc[0], dy = Xp[1] - Xc[1];
        const float r = fmaxf(sqrtf(dx * dx + dy * dy), 1e-8f);
        const float ang = atanf((Xp[2] - zc) / r);
        const float w = confs[(int64_t)k * HW + p] - 0.999f;
        sa += w * ang; sw += w; sr += r;
    }
    const float depth = (sr / (float)n) * tanf(sa / sw);
    canon2[p] = 1.0f + depth / canon[3 * c + 2];
}

// dust3r estimate_focal_knowing_depth(focal_mode='weiszfeld') [U] for ALL views in one launch.
// Pass 0 is the closed form f = mean(xy/z . pixel) / mean(|xy/z|^2); passes 1..10 repeat it with weights 1 / distance.
// Each pass ends in a reduction over the whole image, so a view is spread over G workgroups that meet after every pass
// (round 1 ran one 1024-thread workgroup per view and one launch per view: 1.14 ms x views, all of it dependent
// latency on one CU).  The meeting is a counter barrier per view: lane 0 publishes the workgroup's partial sums as ONE
// 8-byte word (relaxed agent-scope store: write-through, flag-free -- every word of `partials` is written once per
// launch), drains it, bumps the view's counter and polls it; afterwards every workgroup adds the G partials in the
// same fixed order, so all of them continue with the same focal bit for bit (and the result is reproducible).
// All G x views workgroups are resident at once (G is chosen so that there are at most 1024 of them).
#define FB_THREADS 256
#define FB_MAX_G 64

__device__ __forceinline__ float2 block_sum2_fb(float a, float b, float (*red)[FB_THREADS]) {
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = FB_THREADS / 2; s > 0; s >>= 1) {   // fixed-order tree over the workgroup
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    const float2 t = make_float2(red[0][0], red[1][0]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(FB_THREADS) void k_focal_weiszfeld(int n_views, int G, int H, int W,
                                                                const float* __restrict__ canon_all, float ppx,
                                                                float ppy, float min_focal, float max_focal,
                                                                unsigned long long* partials, unsigned* counters,
                                                                float* __restrict__ focal_out) {
    __shared__ float red[2][FB_THREADS];
    __shared__ float s_focal;
    const int view = blockIdx.y, wg = blockIdx.x;
    const int HW = H * W;
    const float* canon = canon_all + (size_t)view * HW * 3;
    float focal = 0.f;
    for (int it = 0; it <= 10; ++it) {
        float num = 0.f, den = 0.f;
        for (int p = wg * FB_THREADS + threadIdx.x; p < HW; p += G * FB_THREADS) {
            const int y = p / W, x = p - y * W;
            const float u = (float)x - ppx, v = (float)y - ppy;
            const float z = canon[3 * p + 2];
            float a = canon[3 * p] / z, b = canon[3 * p + 1] / z;
            if (!(fabsf(a) <= 3.4028234663852886e+38f)) a = 0.f;   // nan_to_num(nan = 0, posinf = 0, neginf = 0)
            if (!(fabsf(b) <= 3.4028234663852886e+38f)) b = 0.f;
            const float dpx = a * u + b * v, dxx = a * a + b * b;
            float w = 1.f;
            if (it > 0) {
                const float eu = u - focal * a, ev = v - focal * b;
                w = 1.0f / fmaxf(sqrtf(eu * eu + ev * ev), 1e-8f);
            }
            num += w * dpx; den += w * dxx;
        }
        const float2 t = block_sum2_fb(num, den, red);
        unsigned long long* slot = partials + ((size_t)it * n_views + view) * G;
        if (threadIdx.x == 0) {
            const unsigned long long word = ((unsigned long long)__float_as_uint(t.y) << 32) | __float_as_uint(t.x);
            __hip_atomic_store(slot + wg, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the word has left before the arrival is counted
            __hip_atomic_fetch_add(counters + view, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(it + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(counters + view, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 24)) __builtin_trap();   // every workgroup of the grid is resident: cannot happen
            }
        }
        __syncthreads();
        // the G partials, added in a fixed order (identical in every workgroup)
        float pn = 0.f, pd = 0.f;
        if ((int)threadIdx.x < G) {
            const unsigned long long word = __hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pn = __uint_as_float((unsigned)word); pd = __uint_as_float((unsigned)(word >> 32));
        }
        const float2 tot = block_sum2_fb(pn, pd, red);
        if (threadIdx.x == 0) s_focal = (tot.x / (float)HW) / (tot.y / (float)HW);
        __syncthreads();
        focal = s_focal;
    }
    if (threadIdx.x == 0 && wg == 0) {
        const float base = (float)max(H, W) / (2.0f * 0.57735026918962576f);   // 2 tan(30 deg)
        focal_out[view] = fminf(fmaxf(focal, min_focal * base), max_focal * base);
    }
}

__global__ __launch_bounds__(256) void k_anchor_offsets(int64_t n, int H, int W, int S,
                                                        const float* __restrict__ canon2,
                                                        const float* __restrict__ xy, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ off_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = (int)xy[2 * i], py = (int)xy[2 * i + 1];     // .long(): truncation
    const int bx = px / S, by = py / S, W2 = (W - S / 2 + S - 1) / S;   // len(range(S/2, W, S))
    idx_out[i] = by * W2 + bx;
    const float ref = canon2[(by * S + S / 2) * W + bx * S + S / 2];
    off_out[i] = canon2[py * W + px] / ref;
}

ST3R_EXPORT int st3r_canon_view(st3r_ctx* ctx, void* stream, int n, int H, int W, int subsample, const float* ptmaps,
                                const float* confs, float* canon, float* canon2, float* cconf) {
    ARG_CHECK(ctx && n > 0 && H > 0 && W > 0 && subsample > 0 && H % subsample == 0 && W % subsample == 0);
    ARG_CHECK(ptmaps && confs && canon && canon2 && cconf);
    const int HW = H * W;
    hipLaunchKernelGGL(k_canon_mean, dim3(ceil_div(HW, 256)), dim3(256), 0, (hipStream_t)stream, n, HW, ptmaps, confs,
                       canon, cconf);
    hipLaunchKernelGGL(k_canon_angle, dim3(ceil_div(HW, 256)), dim3(256), 0, (hipStream_t)stream, n, H, W, subsample,
                       ptmaps, confs, canon, canon2);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_focal_weiszfeld_batch(st3r_ctx* ctx, void* stream, int n_views, int H, int W, const float* canon,
                                           float ppx, float ppy, float min_focal, float max_focal, float* focal_out) {
    ARG_CHECK(ctx && n_views > 0 && n_views <= 1024 && H > 0 && W > 0 && canon && focal_out && min_focal > 0.f &&
              max_focal >= min_focal);
    hipStream_t s = (hipStream_t)stream;
    // G workgroups per view: enough to spread an image over many CUs, few enough that the whole grid is resident
    int G = min(FB_MAX_G, max(1, 1024 / n_views));
    G = min(G, max(1, ceil_div((int64_t)H * W, FB_THREADS)));
    // the per-view counter barrier needs every workgroup of the grid resident at once: bound the grid by what THIS
    // device can hold (a partitioned GPU or a CU mask leaves fewer CUs than the 256 of a whole MI355X); one workgroup
    // per view (G = 1) needs no partner and always works
    {
        int per_cu = 0, cus = 0;
        hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_focal_weiszfeld, FB_THREADS, 0);
        hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const int64_t resident = (e1 == hipSuccess && e2 == hipSuccess) ? (int64_t)per_cu * cus / 2 : 0;   // half: other streams
        if (resident < (int64_t)G * n_views) G = (int)max((int64_t)1, resident / n_views);
    }
    const size_t part_bytes = sizeof(unsigned long long) * 11 * (size_t)n_views * G;
    const size_t cnt_off = (part_bytes + 255) & ~(size_t)255;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_SCAN_TMP, cnt_off + sizeof(unsigned) * (size_t)n_views, &p);
    if (rc) return rc;
    unsigned* counters = (unsigned*)((char*)p + cnt_off);
    HIP_TRY(hipMemsetAsync(counters, 0, sizeof(unsigned) * (size_t)n_views, s));
    hipLaunchKernelGGL(k_focal_weiszfeld, dim3(G, n_views), dim3(FB_THREADS), 0, s, n_views, G, H, W, canon, ppx, ppy,
                       min_focal, max_focal, (unsigned long long*)p, counters, focal_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_focal_weiszfeld(st3r_ctx* ctx, void* stream, int H, int W, const float* canon, float ppx,
                                     float ppy, float min_focal, float max_focal, float* focal_out) {
    return st3r_focal_weiszfeld_batch(ctx, stream, 1, H, W, canon, ppx, ppy, min_focal, max_focal, focal_out);
}

ST3R_EXPORT int st3r_anchor_offsets(st3r_ctx* ctx, void* stream, int64_t n, int H, int W, int subsample,
                                    const float* canon2, const float* xy, int32_t* idx_out, float* off_out) {
    ARG_CHECK(ctx && n >= 0 && H > 0 && W > 0 && subsample > 0 && canon2);
    if (n == 0) return ST3R_OK;
    ARG_CHECK(xy && idx_out && off_out);
    hipLaunchKernelGGL(k_anchor_offsets, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n, H, W, subsample,
                       canon2, xy, idx_out, off_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}
