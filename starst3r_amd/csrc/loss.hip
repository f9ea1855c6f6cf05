// K8: fused L1 + SSIM loss, forward and backward, for C views.
// Replaces torch.nn.functional.l1_loss + torchmetrics StructuralSimilarityIndexMeasure
// (data_range=1: 11x11 gaussian window sigma 1.5, k1=.01, k2=.03, mean over the interior
// (H-10)x(W-10)) and their autograd, starster/gs.py:126-130,153.
//
// Two LDS-tiled separable passes over 32x32 pixel tiles (halo 5):
//   k_ssim_fwd: x,y tile -> 5 windowed moments -> SSIM map; accumulates sum|x-y| and the
//               SSIM sum; writes the three per-pixel derivative maps D = (dS/dmu_x,
//               dS/dE[x^2], dS/dE[xy]) of the interior (zero elsewhere), 9 floats/pixel;
//   k_ssim_bwd: v_x = k_l1*sign(x-y) + k_ss*( G*D0 + 2x G*D1 + y G*D2 )   (G symmetric).
// The window never touches padding for interior outputs, so reflect-padding is not needed.
#include "common.h"

#define TS 32          // output tile
#define HALO 5
#define TIN (TS + 2 * HALO)  // 42
#define KS 11

struct Win { float w[KS]; };

static Win make_window() {
    double g[KS], s = 0;
    for (int i = 0; i < KS; ++i) { double d = (i - HALO) / 1.5; g[i] = exp(-0.5 * d * d); s += g[i]; }
    Win w;
    for (int i = 0; i < KS; ++i) w.w[i] = (float)(g[i] / s);
    return w;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, const float* __restrict__ render,
                                                  const float* __restrict__ gt, Win win,
                                                  double* __restrict__ sums, float* __restrict__ D) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sx = smem;                       // [TIN][TIN*3]
    float* sy = sx + TIN * TIN * 3;         // [TIN][TIN*3]
    float* hp = sy + TIN * TIN * 3;         // [5][TIN][TS]
    __shared__ float red[4];
    const int cam = blockIdx.z;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const float* xr = render + (int64_t)cam * H * W * 3;
    const float* yr = gt + (int64_t)cam * H * W * 3;
    for (int e = threadIdx.x; e < TIN * TIN * 3; e += 256) {
        const int row = e / (TIN * 3), rem = e - row * (TIN * 3);
        const int col = rem / 3;
        const int i = i0 - HALO + row, j = j0 - HALO + col;
        float vx = 0.f, vy = 0.f;
        if (i >= 0 && i < H && j >= 0 && j < W) {
            const int64_t q = ((int64_t)i * W + j) * 3 + (rem - col * 3);
            vx = xr[q]; vy = yr[q];
        }
        sx[e] = vx; sy[e] = vy;
    }
    __syncthreads();
    // L1 partial over the centre 32x32 of the tile (pixels outside the image hold 0,0)
    float l1 = 0.f;
    for (int e = threadIdx.x; e < TS * TS * 3; e += 256) {
        const int row = e / (TS * 3), rem = e - row * (TS * 3);
        const int a = (row + HALO) * (TIN * 3) + HALO * 3 + rem;
        l1 += fabsf(sy[a] - sx[a]);
    }
    float ssim_acc = 0.f;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    for (int ch = 0; ch < 3; ++ch) {
        // horizontal pass: [TIN rows][TS cols]
        for (int e = threadIdx.x; e < TIN * TS; e += 256) {
            const int row = e / TS, col = e - row * TS;
            const float* px = sx + row * (TIN * 3) + col * 3 + ch;
            const float* py = sy + row * (TIN * 3) + col * 3 + ch;
            float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float x = px[k * 3], y = py[k * 3], w = win.w[k];
                const float wx = w * x, wy = w * y;
                s0 += wx; s1 += wy; s2 += wx * x; s3 += wy * y; s4 += wx * y;
            }
            hp[0 * TIN * TS + e] = s0; hp[1 * TIN * TS + e] = s1; hp[2 * TIN * TS + e] = s2;
            hp[3 * TIN * TS + e] = s3; hp[4 * TIN * TS + e] = s4;
        }
        __syncthreads();
        // vertical pass + SSIM
        for (int e = threadIdx.x; e < TS * TS; e += 256) {
            const int row = e / TS, col = e - row * TS;
            const int i = i0 + row, j = j0 + col;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            const bool interior = (i >= HALO) && (i < H - HALO) && (j >= HALO) && (j < W - HALO);
            if (interior) {
                float mx = 0, my = 0, exx = 0, eyy = 0, exy = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int a = (row + k) * TS + col;
                    const float w = win.w[k];
                    mx += w * hp[0 * TIN * TS + a]; my += w * hp[1 * TIN * TS + a];
                    exx += w * hp[2 * TIN * TS + a]; eyy += w * hp[3 * TIN * TS + a];
                    exy += w * hp[4 * TIN * TS + a];
                }
                const float sxx = exx - mx * mx, syy = eyy - my * my, sxy = exy - mx * my;
                const float n1 = 2.f * mx * my + c1, n2 = 2.f * sxy + c2;
                const float dd1 = mx * mx + my * my + c1, dd2 = sxx + syy + c2;
                const float inv = 1.0f / (dd1 * dd2);
                const float ssim = n1 * n2 * inv;
                ssim_acc += ssim;
                const float dn1 = n2 * inv, dn2 = n1 * inv;
                const float g1 = -ssim / dd1, g2 = -ssim / dd2;
                d0 = 2.f * my * (dn1 - dn2) + 2.f * mx * (g1 - g2);  // dS/dmu_x
                d1 = g2;                                             // dS/dE[x^2]
                d2 = 2.f * dn2;                                      // dS/dE[xy]
            }
            if (D && i < H && j < W) {
                float* dst = D + ((((int64_t)cam * H + i) * W + j) * 3 + ch) * 3;
                dst[0] = d0; dst[1] = d1; dst[2] = d2;
            }
        }
        __syncthreads();
    }
    const float tl1 = block_sum_256(l1, red);
    const float tss = block_sum_256(ssim_acc, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * cam + 0], (double)tl1);
        atomicAdd(&sums[2 * cam + 1], (double)tss);
    }
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, const float* __restrict__ render,
                                                  const float* __restrict__ gt, const float* __restrict__ D,
                                                  Win win, float k_l1, float k_ss,
                                                  float* __restrict__ v_render) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sd = smem;                   // [TIN][TIN*9]
    float* hp = sd + TIN * TIN * 9;     // [3][TIN][TS]
    const int cam = blockIdx.z;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const float* Dc = D + (int64_t)cam * H * W * 9;
    for (int e = threadIdx.x; e < TIN * TIN * 9; e += 256) {
        const int row = e / (TIN * 9), rem = e - row * (TIN * 9);
        const int col = rem / 9;
        const int i = i0 - HALO + row, j = j0 - HALO + col;
        float v = 0.f;
        if (i >= 0 && i < H && j >= 0 && j < W) v = Dc[((int64_t)i * W + j) * 9 + (rem - col * 9)];
        sd[e] = v;
    }
    __syncthreads();
    for (int ch = 0; ch < 3; ++ch) {
        for (int e = threadIdx.x; e < TIN * TS; e += 256) {
            const int row = e / TS, col = e - row * TS;
            const float* pd = sd + row * (TIN * 9) + col * 9 + ch * 3;
            float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float w = win.w[k];
                s0 += w * pd[k * 9]; s1 += w * pd[k * 9 + 1]; s2 += w * pd[k * 9 + 2];
            }
            hp[0 * TIN * TS + e] = s0; hp[1 * TIN * TS + e] = s1; hp[2 * TIN * TS + e] = s2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TS * TS; e += 256) {
            const int row = e / TS, col = e - row * TS;
            const int i = i0 + row, j = j0 + col;
            if (i < H && j < W) {
                float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int a = (row + k) * TS + col;
                    const float w = win.w[k];
                    a0 += w * hp[0 * TIN * TS + a]; a1 += w * hp[1 * TIN * TS + a]; a2 += w * hp[2 * TIN * TS + a];
                }
                const int64_t q = (((int64_t)cam * H + i) * W + j) * 3 + ch;
                const float x = render[q], y = gt[q];
                const float sgn = (x > y) ? 1.0f : ((x < y) ? -1.0f : 0.0f);
                v_render[q] = k_l1 * sgn + k_ss * (a0 + 2.f * x * a1 + y * a2);
            }
        }
        __syncthreads();
    }
}

int st3r_loss_impl(st3r_ctx* ctx, hipStream_t s, int C, int H, int W, const float* render, const float* gt,
                   float w_l1, float w_ssim, double* sums, float* v_render) {
    static const Win win = make_window();
    HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)C, s));
    float* D = nullptr;
    if (v_render) {
        void* p;
        int rc = st3r_arena_get(ctx, SLOT_SSIM_A, sizeof(float) * 9 * (size_t)C * H * W, &p);
        if (rc) return rc;
        D = (float*)p;
    }
    dim3 grid(ceil_div(W, TS), ceil_div(H, TS), C);
    const size_t sh_f = sizeof(float) * (2 * TIN * TIN * 3 + 5 * TIN * TS);
    const size_t sh_b = sizeof(float) * (TIN * TIN * 9 + 3 * TIN * TS);
    static bool attr_set = false;
    if (!attr_set) {  // both kernels need more than the default 64 KiB of dynamic LDS
        HIP_TRY(hipFuncSetAttribute((const void*)k_ssim_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_b));
        HIP_TRY(hipFuncSetAttribute((const void*)k_ssim_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_f));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(256), sh_f, s, H, W, render, gt, win, sums, D);
    LAUNCH_CHECK();
    if (v_render) {
        const int Hi = H - 2 * HALO, Wi = W - 2 * HALO;
        const double cnt = (Hi > 0 && Wi > 0) ? (double)Hi * Wi * 3 : 0.0;
        const float k_l1 = (float)((double)w_l1 / ((double)H * W * 3));
        const float k_ss = cnt > 0 ? (float)(-(double)w_ssim / cnt) : 0.f;
        hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), sh_b, s, H, W, render, gt, D, win, k_l1, k_ss, v_render);
        LAUNCH_CHECK();
    }
    return ST3R_OK;
}

ST3R_EXPORT int st3r_loss_l1_ssim(st3r_ctx* ctx, void* stream, int C, int height, int width, const float* render,
                                  const float* gt, float w_l1, float w_ssim, double* sums, float* v_render) {
    ARG_CHECK(ctx && C > 0 && height > 0 && width > 0 && render && gt && sums);
    return st3r_loss_impl(ctx, (hipStream_t)stream, C, height, width, render, gt, w_l1, w_ssim, sums, v_render);
}
