// K1: projection + degree-1 SH colour + tile count, one thread per Gaussian looping over
// the cameras in registers (each Gaussian's 23 parameter floats are read from HBM once,
// its world covariance is built once and reused for every view).
//
// Replaces gsplat fully_fused_projection (packed) + spherical_harmonics + isect_tiles
// pass 1 as reached from starster/gs.py:76-87.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: radii, tile rectangles and depth keys are
// integer outputs that must be bit-exact against oracle/gs_oracle.c, so every float
// operation below is a single IEEE binary32 op in a fixed order (sums of three products
// left to right; 1/x and sqrt correctly rounded -- hipcc's default for HIP).
#include "common.h"
#include "tile_rect.h"

#define CAM_STRIDE 32
// per-camera block in LDS: [0..11] view matrix rows 0..2 (R|t), 12 fx, 13 fy, 14 cx, 15 cy,
// 16 lim_x_pos, 17 lim_x_neg, 18 lim_y_pos, 19 lim_y_neg, 20..22 camera position

__device__ __forceinline__ float dot3f(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

#define SH_C0 0.2820947917738781f
#define SH_C1 0.48860251190292f

__global__ __launch_bounds__(256) void k_project_sh_fwd(
    int N, int C, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ sh,
    int sh_stride, const float* __restrict__ viewmats, const float* __restrict__ Ks,
    const float* __restrict__ campos, int W, int H, int tile_size, int tile_w, int tile_h, float eps2d,
    float near_plane, float far_plane, float radius_clip, float4* __restrict__ splats,
    int32_t* __restrict__ tiles_per_gauss, double* __restrict__ reg_part,
    uint64_t* __restrict__ depth_keys, int32_t* __restrict__ depth_vals, int tight, uint32_t key_base,
    void* __restrict__ rects, int rect32, uint32_t* __restrict__ krange_part) {
    // krange_part != NULL (fused training calls, <= 8 views): the level-1 keys carry NO camera bits -- the sort treats every
    // camera's N pairs as a segment of its own (radix_sort.hip) -- and the block's smallest / largest key of a visible pair
    // is left for k_reg_reduce, which turns them into the bias and the pass count of that sort
    extern __shared__ float cam[];
    __shared__ float red[8];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float* o = cam + c * CAM_STRIDE;
        const float* V = viewmats + 16 * c;
        for (int k = 0; k < 12; ++k) o[k] = V[k];
        const float* K = Ks + 9 * c;
        float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
        o[12] = fx; o[13] = fy; o[14] = cx; o[15] = cy;
        float tan_fovx = 0.5f * (float)W / fx;
        float tan_fovy = 0.5f * (float)H / fy;
        o[16] = ((float)W - cx) / fx + 0.3f * tan_fovx;
        o[17] = cx / fx + 0.3f * tan_fovx;
        o[18] = ((float)H - cy) / fy + 0.3f * tan_fovy;
        o[19] = cy / fy + 0.3f * tan_fovy;
        o[20] = campos[3 * c]; o[21] = campos[3 * c + 1]; o[22] = campos[3 * c + 2];
    }
    __syncthreads();

    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = g < N;
    float mx = 0, my = 0, mz = 0, opac = 0;
    float cov0 = 0, cov1 = 0, cov2 = 0, cov3 = 0, cov4 = 0, cov5 = 0;
    float k[12];
    float reg_o = 0.f, reg_s = 0.f;
    if (active) {
        mx = means[3 * g]; my = means[3 * g + 1]; mz = means[3 * g + 2];
        opac = opacities[g];
        float qw = quats[4 * g], qx = quats[4 * g + 1], qy = quats[4 * g + 2], qz = quats[4 * g + 3];
        float s0 = scales[3 * g], s1 = scales[3 * g + 1], s2 = scales[3 * g + 2];
        const float* kp = sh + (int64_t)g * sh_stride;
#pragma unroll
        for (int i = 0; i < 12; ++i) k[i] = kp[i];
        float n2 = ((qw * qw + qx * qx) + qy * qy) + qz * qz;
        float inv = 1.0f / sqrtf(n2);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
        float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        float m0 = (1.0f - 2.0f * (y2 + z2)) * s0, m1 = (2.0f * (xy - wz)) * s1, m2 = (2.0f * (xz + wy)) * s2;
        float m3 = (2.0f * (xy + wz)) * s0, m4 = (1.0f - 2.0f * (x2 + z2)) * s1, m5 = (2.0f * (yz - wx)) * s2;
        float m6 = (2.0f * (xz - wy)) * s0, m7 = (2.0f * (yz + wx)) * s1, m8 = (1.0f - 2.0f * (x2 + y2)) * s2;
        cov0 = dot3f(m0, m1, m2, m0, m1, m2);
        cov1 = dot3f(m0, m1, m2, m3, m4, m5);
        cov2 = dot3f(m0, m1, m2, m6, m7, m8);
        cov3 = dot3f(m3, m4, m5, m3, m4, m5);
        cov4 = dot3f(m3, m4, m5, m6, m7, m8);
        cov5 = dot3f(m6, m7, m8, m6, m7, m8);
        if (reg_part) {
            reg_o = 1.0f / (1.0f + __expf(-opac));
            reg_s = (__expf(s0) + __expf(s1)) + __expf(s2);
        }
    }
    if (reg_part) {  // block reduction of the two regulariser sums
        for (int off = 32; off > 0; off >>= 1) {
            reg_o += __shfl_down(reg_o, off);
            reg_s += __shfl_down(reg_s, off);
        }
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[w] = reg_o; red[4 + w] = reg_s; }
        __syncthreads();
        if (threadIdx.x == 0) {
            reg_part[4 * blockIdx.x + 0] = (double)((red[0] + red[1]) + (red[2] + red[3]));
            reg_part[4 * blockIdx.x + 1] = (double)((red[4] + red[5]) + (red[6] + red[7]));
        }
    }
    int n_vis = 0, n_ref = 0;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;   // level-1 keys of this thread's visible pairs
    for (int c = 0; active && c < C; ++c) {
        const float* o = cam + c * CAM_STRIDE;
        const float R00 = o[0], R01 = o[1], R02 = o[2], t0 = o[3];
        const float R10 = o[4], R11 = o[5], R12 = o[6], t1 = o[7];
        const float R20 = o[8], R21 = o[9], R22 = o[10], t2 = o[11];
        const int64_t pid = (int64_t)c * N + g;
        float x = dot3f(R00, R01, R02, mx, my, mz) + t0;
        float y = dot3f(R10, R11, R12, mx, my, mz) + t1;
        float z = dot3f(R20, R21, R22, mx, my, mz) + t2;
        bool valid = !(z < near_plane || z > far_plane);
        float m2x = 0, m2y = 0, ca = 0, cb = 0, cc = 0, radius = 0;
        if (valid) {
            // T = R * cov ; covc = T * R^T (upper)
            float T0 = dot3f(R00, R01, R02, cov0, cov1, cov2), T1 = dot3f(R00, R01, R02, cov1, cov3, cov4),
                  T2 = dot3f(R00, R01, R02, cov2, cov4, cov5);
            float T3 = dot3f(R10, R11, R12, cov0, cov1, cov2), T4 = dot3f(R10, R11, R12, cov1, cov3, cov4),
                  T5 = dot3f(R10, R11, R12, cov2, cov4, cov5);
            float T6 = dot3f(R20, R21, R22, cov0, cov1, cov2), T7 = dot3f(R20, R21, R22, cov1, cov3, cov4),
                  T8 = dot3f(R20, R21, R22, cov2, cov4, cov5);
            float c0 = dot3f(T0, T1, T2, R00, R01, R02);
            float c1 = dot3f(T0, T1, T2, R10, R11, R12);
            float c2 = dot3f(T0, T1, T2, R20, R21, R22);
            float c3 = dot3f(T3, T4, T5, R10, R11, R12);
            float c4 = dot3f(T3, T4, T5, R20, R21, R22);
            float c5 = dot3f(T6, T7, T8, R20, R21, R22);
            const float fx = o[12], fy = o[13], cx = o[14], cy = o[15];
            float rz = 1.0f / z;
            float rz2 = rz * rz;
            float xr = x * rz, yr = y * rz;
            float tx = z * fminf(o[16], fmaxf(-o[17], xr));
            float ty = z * fminf(o[18], fmaxf(-o[19], yr));
            float a = fx * rz, cj = -(fx * tx) * rz2;
            float b = fy * rz, d = -(fy * ty) * rz2;
            float t0x = a * c0 + cj * c2;
            float t0y = a * c1 + cj * c4;
            float t0z = a * c2 + cj * c5;
            float t1y = b * c3 + d * c4;
            float t1z = b * c4 + d * c5;
            float c00 = t0x * a + t0z * cj;
            float c01 = t0y * b + t0z * d;
            float c11 = t1y * b + t1z * d;
            m2x = (fx * x) * rz + cx;
            m2y = (fy * y) * rz + cy;
            c00 += eps2d; c11 += eps2d;
            float det = c00 * c11 - c01 * c01;
            if (det <= 0.0f) valid = false;
            else {
                float inv_det = 1.0f / det;
                ca = c11 * inv_det; cb = -c01 * inv_det; cc = c00 * inv_det;
                float bb = 0.5f * (c00 + c11);
                float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
                radius = ceilf(3.0f * sqrtf(v1));
                if (radius <= radius_clip) valid = false;
                else if (m2x + radius <= 0.0f || m2x - radius >= (float)W || m2y + radius <= 0.0f ||
                         m2y - radius >= (float)H)
                    valid = false;
            }
        }
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
        int ntiles = 0;
        TileRect tr = {0, 0, 0, 0};
        if (valid) {
            // SH colour (degree 1) + clamp_min(c + 0.5, 0)
            float dx = mx - o[20], dy = my - o[21], dz = mz - o[22];
            float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float col[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float r = SH_C0 * k[ch];
                r = r + SH_C1 * ((-dy * k[3 + ch] + dz * k[6 + ch]) - dx * k[9 + ch]);
                r = r + 0.5f;
                col[ch] = r < 0.0f ? 0.0f : r;
            }
            // tile rectangle (isect_tiles pass 1)
            tr = ref_tile_rect(m2x, m2y, radius, tile_size, tile_w, tile_h);
            n_ref += (tr.y1 - tr.y0) * (tr.x1 - tr.x0);
            if (tight) tr = tight_tile_rect(tr, m2x, m2y, opac, ca, cb, cc);  // fused train path only
            ntiles = (tr.y1 - tr.y0) * (tr.x1 - tr.x0);
            r0 = make_float4(m2x, m2y, opac, ca);
            r1 = make_float4(cb, cc, col[0], col[1]);
            r2 = make_float4(col[2], z, __int_as_float((int)radius), 0.0f);
        }
        splats[pid * 3 + 0] = r0;
        splats[pid * 3 + 1] = r1;
        splats[pid * 3 + 2] = r2;
        if (tiles_per_gauss) tiles_per_gauss[pid] = ntiles;
        if (rects) rect_store(rects, rect32, pid, tr);
        if (depth_keys) {  // (camera | depth bits) key of the two-level sort; culled pairs sort last
            const uint32_t dbits = valid ? (uint32_t)__float_as_int(z) : 0xFFFFFFFFu;
            if (key_base) {  // packed 32-bit key: camera (<= 8) | depth bits above those of the near plane (29 bits)
                const uint32_t kk = valid ? dbits - key_base : 0x1FFFFFFFu;
                reinterpret_cast<uint32_t*>(depth_keys)[pid] = (krange_part ? 0u : ((uint32_t)c << 29)) | kk;
                if (valid) { kmin = min(kmin, kk); kmax = max(kmax, kk); }
            } else
                depth_keys[pid] = ((uint64_t)c << 32) | dbits;
            if (!krange_part) depth_vals[pid] = (int32_t)pid;   // (the segmented sort's first pass numbers the pairs itself)
        }
        n_vis += valid ? 1 : 0;
    }
    if (krange_part) {
        for (int off = 32; off > 0; off >>= 1) { kmin = min(kmin, (uint32_t)__shfl_down((int)kmin, off)); kmax = max(kmax, (uint32_t)__shfl_down((int)kmax, off)); }
        __shared__ uint32_t redk[8];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { redk[w] = kmin; redk[4 + w] = kmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
            krange_part[2 * blockIdx.x + 0] = min(min(redk[0], redk[1]), min(redk[2], redk[3]));
            krange_part[2 * blockIdx.x + 1] = max(max(redk[4], redk[5]), max(redk[6], redk[7]));
        }
    }
    if (reg_part) {  // visible (camera, gaussian) pairs and reference tile intersections of this block
        for (int off = 32; off > 0; off >>= 1) { n_vis += __shfl_down(n_vis, off); n_ref += __shfl_down(n_ref, off); }
        __shared__ int redi[8];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { redi[w] = n_vis; redi[4 + w] = n_ref; }
        __syncthreads();
        if (threadIdx.x == 0) {
            reg_part[4 * blockIdx.x + 2] = (double)((redi[0] + redi[1]) + (redi[2] + redi[3]));
            reg_part[4 * blockIdx.x + 3] = (double)((redi[4] + redi[5]) + (redi[6] + redi[7]));
        }
    }
}

// reg_sums[k] += (or =, `overwrite`) sum over blocks of reg_part[., k], in a fixed order (bit-reproducible; thousands of
// same-address double atomics from the projection kernel used to cost more than the projection itself).  The fused
// training calls also let this one small workgroup clear the loss kernel's accumulators (zero_ptr[0 .. zero_n)): two
// memset launches less per step.
// krange (with krange_part, the fused training calls): [0] = smallest level-1 key of a visible pair, [1] = the key the
// culled pairs get after the bias (largest - smallest + 1: they sort last), [2] = the 8-bit passes the segmented level-1
// sort needs for keys up to [1] -- three when the depth codes of the scene span less than 2^24 (SYNTH-1M: 2^23.3), else four.
__global__ __launch_bounds__(256) void k_reg_reduce(int n_blocks, const double* __restrict__ reg_part,
                                                    double* __restrict__ reg_sums, int overwrite,
                                                    double* __restrict__ zero_ptr, int zero_n,
                                                    const uint32_t* __restrict__ krange_part,
                                                    uint32_t* __restrict__ krange) {
    for (int i = threadIdx.x; i < zero_n; i += 256) zero_ptr[i] = 0.0;
    if (krange_part) {
        __shared__ uint32_t shk[2][256];
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int b = threadIdx.x; b < n_blocks; b += 256) { lo = min(lo, krange_part[2 * b]); hi = max(hi, krange_part[2 * b + 1]); }
        shk[0][threadIdx.x] = lo; shk[1][threadIdx.x] = hi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) {
                shk[0][threadIdx.x] = min(shk[0][threadIdx.x], shk[0][threadIdx.x + off]);
                shk[1][threadIdx.x] = max(shk[1][threadIdx.x], shk[1][threadIdx.x + off]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            lo = shk[0][0]; hi = shk[1][0];
            if (hi < lo) { lo = 0u; hi = 0u; }             // nothing visible
            const uint32_t top = hi - lo + 1u;            // the culled pairs' key; < 2^29
            krange[0] = lo; krange[1] = top;
            krange[2] = top < (1u << 8) ? 1u : top < (1u << 16) ? 2u : top < (1u << 24) ? 3u : 4u;
        }
        __syncthreads();
    }
    __shared__ double sh[4][256];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += 256)
        for (int k = 0; k < 4; ++k) acc[k] += reg_part[4 * b + k];
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) reg_sums[threadIdx.x] = (overwrite ? 0.0 : reg_sums[threadIdx.x]) + sh[threadIdx.x][0];
}

// depth_keys/depth_vals (optional, [C*N]): per-pair (camera | depth bits) key and pair id for the
// two-level sort of the fused path
int st3r_project_impl(st3r_ctx* ctx, hipStream_t s, int N, int C, const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* sh, int sh_stride, const float* viewmats, const float* Ks,
                      const float* campos, int width, int height, int tile_size, float eps2d, float near_plane,
                      float far_plane, float radius_clip, float* splats, int32_t* tiles_per_gauss, double* reg_sums,
                      uint64_t* depth_keys, int32_t* depth_vals, int tight, uint32_t key_base, void* rects, int rect32,
                      int reg_overwrite, double* zero_ptr, int zero_n, uint32_t* krange) {
    // krange != NULL (needs reg_sums): segment keys + key range for the segmented level-1 sort (see k_reg_reduce)
    if (N == 0) {
        if (reg_sums && reg_overwrite) HIP_TRY(hipMemsetAsync(reg_sums, 0, sizeof(double) * 4, s));
        if (zero_ptr && zero_n > 0) HIP_TRY(hipMemsetAsync(zero_ptr, 0, sizeof(double) * (size_t)zero_n, s));
        return ST3R_OK;
    }
    int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
    dim3 grid(ceil_div(N, 256)), block(256);
    size_t shmem = (size_t)C * CAM_STRIDE * sizeof(float);
    double* reg_part = nullptr;
    if (reg_sums) {
        void* p;
        int rc = st3r_arena_get(ctx, SLOT_REG_PART, sizeof(double) * 4 * (size_t)grid.x, &p);
        if (rc) return rc;
        reg_part = (double*)p;
    }
    uint32_t* krange_part = nullptr;
    if (krange && reg_sums) {
        void* p;
        int rc = st3r_arena_get(ctx, SLOT_KRANGE_PART, sizeof(uint32_t) * 2 * (size_t)grid.x, &p);
        if (rc) return rc;
        krange_part = (uint32_t*)p;
    }
    hipLaunchKernelGGL(k_project_sh_fwd, grid, block, shmem, s, N, C, means, quats, scales, opacities, sh, sh_stride,
                       viewmats, Ks, campos, width, height, tile_size, tile_w, tile_h, eps2d, near_plane, far_plane,
                       radius_clip, (float4*)splats, tiles_per_gauss, reg_part, depth_keys, depth_vals, tight, key_base,
                       rects, rect32, krange_part);
    if (reg_sums)
        hipLaunchKernelGGL(k_reg_reduce, dim3(1), dim3(256), 0, s, (int)grid.x, reg_part, reg_sums, reg_overwrite, zero_ptr,
                           zero_ptr ? zero_n : 0, (const uint32_t*)krange_part, krange);
    else if (zero_ptr && zero_n > 0)
        HIP_TRY(hipMemsetAsync(zero_ptr, 0, sizeof(double) * (size_t)zero_n, s));
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_project_sh(st3r_ctx* ctx, void* stream, int N, int C, const float* means,
                                   const float* quats, const float* scales, const float* opacities,
                                   const float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                   const float* campos, int width, int height, int tile_size, float eps2d,
                                   float near_plane, float far_plane, float radius_clip, float* splats,
                                   int32_t* tiles_per_gauss, double* reg_sums) {
    ARG_CHECK(ctx && N >= 0 && C > 0 && C <= ST3R_MAX_VIEWS && sh_stride >= 12 && width > 0 && height > 0 && tile_size > 0);
    ARG_CHECK(means && quats && scales && opacities && sh && viewmats && Ks && campos && splats && tiles_per_gauss);
    return st3r_project_impl(ctx, (hipStream_t)stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks,
                             campos, width, height, tile_size, eps2d, near_plane, far_plane, radius_clip, splats,
                             tiles_per_gauss, reg_sums, nullptr, nullptr, 0, 0u, nullptr, 0, 0, nullptr, 0, nullptr);
}
