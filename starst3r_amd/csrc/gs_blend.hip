// K6/K7: per-tile alpha blending, forward and backward.
// Replaces gsplat rasterize_to_pixels fwd/bwd (starster/gs.py:76 and the autograd pass of
// starster/gs.py:153).
//
// One workgroup = one 16x16 tile = 4 wave64; wave w owns the 8x8 quadrant (w&1, w>>1) so a
// wave covers a compact pixel block (better chance that a whole wave is skipped or finishes
// early than with 16x4 strips).  The tile's depth-sorted Gaussian list is staged through
// LDS in batches of 256 splat records (one record per thread, 3 x 16-byte loads), and every
// lane then reads the same LDS address (broadcast, conflict free).
// Workgroups are remapped so that each XCD (private 4 MiB L2) walks a contiguous range of
// (camera, tile) ids: with 8 views on 8 XCDs every XCD owns one camera's splat array.
#include "common.h"

#define BLK 256

__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ void tile_pixel(int tid, int& lx, int& ly) {
    const int w = tid >> 6, lane = tid & 63;
    lx = ((w & 1) << 3) + (lane & 7);
    ly = ((w >> 1) << 3) + (lane >> 3);
}

__global__ __launch_bounds__(BLK) void k_blend_fwd(int C, int W, int H, int tile_w, int tile_h,
                                                   const float4* __restrict__ splats,
                                                   const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ flat, int n_isects,
                                                   float* __restrict__ out_rgb, float* __restrict__ out_alpha,
                                                   int32_t* __restrict__ last_ids) {
    __shared__ float4 sA[BLK];  // x y opacity conic.a
    __shared__ float4 sB[BLK];  // conic.b conic.c r g
    __shared__ float sC[BLK];   // b
    const int n_tiles = tile_w * tile_h, total = C * n_tiles;
    const int lb = xcd_remap(blockIdx.x, total);
    const int cam = lb / n_tiles, tile = lb - cam * n_tiles;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    int lx, ly;
    tile_pixel(threadIdx.x, lx, ly);
    const int i = ty * 16 + ly, j = tx * 16 + lx;
    const bool inside = (i < H) && (j < W);
    bool done = !inside;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int start = offsets[lb];
    const int end = (lb == total - 1) ? n_isects : offsets[lb + 1];
    float T = 1.0f, r = 0.f, g = 0.f, b = 0.f;
    int cur = 0;
    for (int bs = start; bs < end; bs += BLK) {
        if (__syncthreads_and(done)) break;
        const int idx = bs + threadIdx.x;
        if (idx < end) {
            const int64_t id = flat[idx];
            const float4 a = splats[id * 3 + 0];
            const float4 bq = splats[id * 3 + 1];
            const float4 c = splats[id * 3 + 2];
            sA[threadIdx.x] = a; sB[threadIdx.x] = bq; sC[threadIdx.x] = c.x;
        }
        __syncthreads();
        const int bsz = min(BLK, end - bs);
        for (int t = 0; t < bsz && !done; ++t) {
            const float4 a = sA[t];
            const float4 bq = sB[t];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = 0.5f * (a.w * dx * dx + bq.y * dy * dy) + bq.x * dx * dy;
            const float alpha = fminf(0.999f, a.z * __expf(-sigma));
            if (sigma < 0.f || alpha < 1.f / 255.f) continue;
            const float nT = T * (1.0f - alpha);
            if (nT <= 1e-4f) { done = true; break; }
            const float vis = alpha * T;
            r += bq.z * vis; g += bq.w * vis; b += sC[t] * vis;
            cur = bs + t;
            T = nT;
        }
    }
    if (inside) {
        const int64_t p = ((int64_t)cam * H + i) * W + j;
        out_rgb[3 * p] = r; out_rgb[3 * p + 1] = g; out_rgb[3 * p + 2] = b;
        out_alpha[p] = 1.0f - T;
        last_ids[p] = cur;
    }
}

int st3r_blend_fwd_impl(hipStream_t s, int C, int W, int H, int tile_w, int tile_h, const float* splats,
                        const int32_t* offsets, const int32_t* flat, int64_t n_isects, float* rgb, float* alpha,
                        int32_t* last_ids) {
    const int total = C * tile_w * tile_h;
    hipLaunchKernelGGL(k_blend_fwd, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h, (const float4*)splats,
                       offsets, flat, (int)n_isects, rgb, alpha, last_ids);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_blend_fwd(st3r_ctx* ctx, void* stream, int C, int width, int height, int tile_size,
                                  int tile_w, int tile_h, const float* splats, const int32_t* offsets,
                                  const int32_t* flatten_ids, int64_t n_isects, float* rgb, float* alpha,
                                  int32_t* last_ids) {
    ARG_CHECK(ctx && C > 0 && width > 0 && height > 0 && tile_size == 16);
    ARG_CHECK(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16);
    ARG_CHECK(splats && offsets && rgb && alpha && last_ids && n_isects >= 0 && n_isects < 2147483647LL);
    ARG_CHECK(n_isects == 0 || flatten_ids);
    return st3r_blend_fwd_impl((hipStream_t)stream, C, width, height, tile_w, tile_h, splats, offsets, flatten_ids,
                               n_isects, rgb, alpha, last_ids);
}

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over the 64 lanes of a wave; result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

#define ACC_STRIDE 9

__global__ __launch_bounds__(BLK) void k_blend_bwd(int C, int W, int H, int tile_w, int tile_h,
                                                   const float4* __restrict__ splats,
                                                   const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ flat, int n_isects,
                                                   const float* __restrict__ out_alpha,
                                                   const int32_t* __restrict__ last_ids,
                                                   const float* __restrict__ v_rgb,
                                                   const float* __restrict__ v_alpha,
                                                   float* __restrict__ v_splats) {
    __shared__ float4 sA[BLK];
    __shared__ float4 sB[BLK];
    __shared__ float sC[BLK];
    __shared__ int sId[BLK];
    __shared__ float sAcc[BLK * ACC_STRIDE];  // per-batch, per-Gaussian partial sums of the 4 waves
    const int n_tiles = tile_w * tile_h, total = C * n_tiles;
    const int lb = xcd_remap(blockIdx.x, total);
    const int cam = lb / n_tiles, tile = lb - cam * n_tiles;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    int lx, ly;
    tile_pixel(threadIdx.x, lx, ly);
    const int i = ty * 16 + ly, j = tx * 16 + lx;
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int start = offsets[lb];
    const int end = (lb == total - 1) ? n_isects : offsets[lb + 1];
    if (end <= start) return;
    const int64_t p = ((int64_t)cam * H + i) * W + j;
    float T_final = 1.0f, vr = 0.f, vg = 0.f, vb = 0.f, va = 0.f;
    int bin_final = -1;
    if (inside) {
        T_final = 1.0f - out_alpha[p];
        vr = v_rgb[3 * p]; vg = v_rgb[3 * p + 1]; vb = v_rgb[3 * p + 2];
        if (v_alpha) va = v_alpha[p];
        bin_final = last_ids[p];
    }
    float T = T_final;
    float bufr = 0.f, bufg = 0.f, bufb = 0.f;
    const int wave_bin_final = wave_max_i(bin_final);
    const int lane = threadIdx.x & 63;

    // batches walk the tile list back to front; batch_end is the last index of the batch
    for (int batch_end = end - 1; batch_end >= start; batch_end -= BLK) {
        __syncthreads();
        const int bsz = min(BLK, batch_end + 1 - start);
        const int idx = batch_end - threadIdx.x;
        if (idx >= start) {
            const int64_t id = flat[idx];
            const float4 a = splats[id * 3 + 0];
            const float4 bq = splats[id * 3 + 1];
            const float4 c = splats[id * 3 + 2];
            sA[threadIdx.x] = a; sB[threadIdx.x] = bq; sC[threadIdx.x] = c.x; sId[threadIdx.x] = (int)id;
        }
#pragma unroll
        for (int k = 0; k < ACC_STRIDE; ++k) sAcc[threadIdx.x * ACC_STRIDE + k] = 0.f;
        __syncthreads();
        // t = 0 is the furthest-back Gaussian of the batch (sorted index batch_end - t)
        for (int t = max(0, batch_end - wave_bin_final); t < bsz; ++t) {
            const float4 a = sA[t];
            const float4 bq = sB[t];
            const float cb_ = sC[t];
            bool valid = inside && (batch_end - t <= bin_final);
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = 0.5f * (a.w * dx * dx + bq.y * dy * dy) + bq.x * dx * dy;
            const float vis = __expf(-sigma);
            const float alpha = fminf(0.999f, a.z * vis);
            if (sigma < 0.f || alpha < 1.f / 255.f) valid = false;
            if (!__any(valid)) continue;
            float g_x = 0.f, g_y = 0.f, g_o = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
            if (valid) {
                const float ra = 1.0f / (1.0f - alpha);
                T *= ra;
                const float fac = alpha * T;
                g_r = fac * vr; g_g = fac * vg; g_b = fac * vb;
                float v_al = (bq.z * T - bufr * ra) * vr + (bq.w * T - bufg * ra) * vg + (cb_ * T - bufb * ra) * vb;
                v_al += T_final * ra * va;
                if (a.z * vis <= 0.999f) {
                    const float v_sigma = -a.z * vis * v_al;
                    g_ca = 0.5f * v_sigma * dx * dx;
                    g_cb = v_sigma * dx * dy;
                    g_cc = 0.5f * v_sigma * dy * dy;
                    g_x = v_sigma * (a.w * dx + bq.x * dy);
                    g_y = v_sigma * (bq.x * dx + bq.y * dy);
                    g_o = vis * v_al;
                }
                bufr += bq.z * fac; bufg += bq.w * fac; bufb += cb_ * fac;
            }
            g_x = wave_sum(g_x); g_y = wave_sum(g_y); g_o = wave_sum(g_o);
            g_ca = wave_sum(g_ca); g_cb = wave_sum(g_cb); g_cc = wave_sum(g_cc);
            g_r = wave_sum(g_r); g_g = wave_sum(g_g); g_b = wave_sum(g_b);
            if (lane == 0) {
                float* acc = sAcc + t * ACC_STRIDE;
                atomicAdd(acc + 0, g_x); atomicAdd(acc + 1, g_y); atomicAdd(acc + 2, g_o);
                atomicAdd(acc + 3, g_ca); atomicAdd(acc + 4, g_cb); atomicAdd(acc + 5, g_cc);
                atomicAdd(acc + 6, g_r); atomicAdd(acc + 7, g_g); atomicAdd(acc + 8, g_b);
            }
        }
        __syncthreads();
        if (threadIdx.x < bsz) {
            const float* acc = sAcc + threadIdx.x * ACC_STRIDE;
            bool any = false;
#pragma unroll
            for (int k = 0; k < ACC_STRIDE; ++k) any |= (acc[k] != 0.f);
            if (any) {
                float* dst = v_splats + (int64_t)sId[threadIdx.x] * ST3R_SPLAT_STRIDE;
#pragma unroll
                for (int k = 0; k < ACC_STRIDE; ++k) atomicAdd(dst + k, acc[k]);
            }
        }
    }
}

int st3r_blend_bwd_impl(hipStream_t s, int C, int W, int H, int tile_w, int tile_h, const float* splats,
                        const int32_t* offsets, const int32_t* flat, int64_t n_isects, const float* alpha,
                        const int32_t* last_ids, const float* v_rgb, const float* v_alpha, int64_t n_pairs,
                        float* v_splats) {
    HIP_TRY(hipMemsetAsync(v_splats, 0, sizeof(float) * ST3R_SPLAT_STRIDE * (size_t)n_pairs, s));
    if (n_isects == 0) return ST3R_OK;
    const int total = C * tile_w * tile_h;
    hipLaunchKernelGGL(k_blend_bwd, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h, (const float4*)splats,
                       offsets, flat, (int)n_isects, alpha, last_ids, v_rgb, v_alpha, v_splats);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_blend_bwd(st3r_ctx* ctx, void* stream, int C, int width, int height, int tile_size,
                                  int tile_w, int tile_h, const float* splats, const int32_t* offsets,
                                  const int32_t* flatten_ids, int64_t n_isects, const float* alpha,
                                  const int32_t* last_ids, const float* v_rgb, const float* v_alpha,
                                  int64_t n_pairs, float* v_splats) {
    ARG_CHECK(ctx && C > 0 && width > 0 && height > 0 && tile_size == 16);
    ARG_CHECK(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16);
    ARG_CHECK(splats && offsets && alpha && last_ids && v_rgb && v_splats && n_pairs >= 0);
    ARG_CHECK(n_isects >= 0 && n_isects < 2147483647LL && (n_isects == 0 || flatten_ids));
    return st3r_blend_bwd_impl((hipStream_t)stream, C, width, height, tile_w, tile_h, splats, offsets, flatten_ids,
                               n_isects, alpha, last_ids, v_rgb, v_alpha, n_pairs, v_splats);
}
