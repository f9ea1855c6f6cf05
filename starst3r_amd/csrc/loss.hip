// K8: fused L1 + SSIM loss, forward and backward, for C views.
// Replaces torch.nn.functional.l1_loss + torchmetrics StructuralSimilarityIndexMeasure
// (data_range=1: 11x11 gaussian window sigma 1.5, k1=.01, k2=.03, mean over the interior
// (H-10)x(W-10)) and their autograd, starster/gs.py:126-130,153.
//
// Row-streaming separable convolution.  A workgroup owns a strip 64 pixels wide (x3 interleaved channels, one
// thread per (column, channel)) and walks down its rows plus the halo.  Each row segment is staged once in LDS
// (coalesced: interleaved channels make the segment one contiguous run), the 11-tap horizontal pass reads it from
// LDS, and the vertical pass never touches LDS: every thread keeps the last 11 horizontal results in a register ring
// (the row loop is unrolled by 11 so ring slots are compile-time indices).
//   forward:  x,y -> 5 windowed moments -> SSIM map; accumulates sum|x-y| and the SSIM sum; per pixel the derivative
//             maps D = (dS/dmu_x, dS/dE[x^2], dS/dE[xy]) of the interior (zero elsewhere), 9 floats/pixel;
//   backward: v_x = k_l1*sign(x-y) + k_ss*( G*D0 + 2x G*D1 + y G*D2 )   (G symmetric).
// k_ssim_fwd is the forward alone (value-only calls); k_ssim_fused runs both with D kept in LDS (see there).
// The window never touches padding for interior outputs, so reflect-padding is not needed.
#include "common.h"

#define LW 64                 // strip width in pixels
#define LT (LW * 3)           // threads per workgroup: one per (column, channel)
#define HALO 5
#define KS 11
#define SEG ((LW + 2 * HALO) * 3)   // floats per staged row segment of an image (222)
#define SEGD ((LW + 2 * HALO) * 9)  // floats per staged row segment of D (666)

struct Win { float w[KS]; };

static Win make_window() {
    double g[KS], s = 0;
    for (int i = 0; i < KS; ++i) { double d = (i - HALO) / 1.5; g[i] = exp(-0.5 * d * d); s += g[i]; }
    Win w;
    for (int i = 0; i < KS; ++i) w.w[i] = (float)(g[i] / s);
    return w;
}

__device__ __forceinline__ float block_sum_192(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const float t = (red[0] + red[1]) + red[2];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(LT) void k_ssim_fwd(int LH, int H, int W, const float* __restrict__ render,
                                                 const float* __restrict__ gt, Win win,
                                                 double* __restrict__ sums, float* __restrict__ D) {
    __shared__ float sx[2][SEG];
    __shared__ float sy[2][SEG];
    __shared__ float red[4];
    const int cam = blockIdx.z;
    const int j0 = blockIdx.x * LW, i0 = blockIdx.y * LH;
    const int t = threadIdx.x;
    const int col = t / 3, ch = t - col * 3;
    const int j = j0 + col;
    const float* xr = render + (int64_t)cam * H * W * 3;
    const float* yr = gt + (int64_t)cam * H * W * 3;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const int nrows = min(LH, H - i0) + 2 * HALO;  // input rows i0-5 .. i0+rows+4

    // Input row r (image row i0 - HALO + r) travels HBM -> registers (prefetch, issued two rows
    // ahead of its use so the load latency hides behind a whole row of arithmetic) -> LDS (commit).
    constexpr int NPF = (SEG + LT - 1) / LT;
    float pfx[NPF], pfy[NPF];
    auto prefetch = [&](int r) {
        const int i = i0 - HALO + r;
        const bool row_ok = (r < nrows) && (i >= 0) && (i < H);
#pragma unroll
        for (int n = 0; n < NPF; ++n) {
            const int e = t + n * LT;
            const int jj = j0 - HALO + e / 3;
            float vx = 0.f, vy = 0.f;
            if (row_ok && e < SEG && jj >= 0 && jj < W) {
                const int64_t q = ((int64_t)i * W + j0 - HALO) * 3 + e;
                vx = xr[q]; vy = yr[q];
            }
            pfx[n] = vx; pfy[n] = vy;
        }
    };
    auto commit = [&](int b) {
#pragma unroll
        for (int n = 0; n < NPF; ++n) {
            const int e = t + n * LT;
            if (e < SEG) { sx[b][e] = pfx[n]; sy[b][e] = pfy[n]; }
        }
    };

    float ring[KS][5];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < 5; ++m) ring[s][m] = 0.f;
    float l1 = 0.f, ssim_acc = 0.f;

    prefetch(0);
    commit(0);
    prefetch(1);
    __syncthreads();
    for (int rb = 0; rb < nrows; rb += KS) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int r = rb + s;
            if (r < nrows) {
                const int b = r & 1;
                commit(b ^ 1);     // row r+1 (prefetched one iteration ago)
                prefetch(r + 2);
                // horizontal pass of input row r for this (column, channel)
                const float* px = &sx[b][col * 3 + ch];
                const float* py = &sy[b][col * 3 + ch];
                float h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const float x = px[k * 3], y = py[k * 3], w = win.w[k];
                    const float wx = w * x, wy = w * y;
                    h0 += wx; h1 += wy; h2 += wx * x; h3 += wy * y; h4 += wx * y;
                }
                ring[s][0] = h0; ring[s][1] = h1; ring[s][2] = h2; ring[s][3] = h3; ring[s][4] = h4;
                const int i = i0 - HALO + r;  // image row just staged
                if (i >= i0 && i < i0 + LH && i < H && j < W) l1 += fabsf(py[HALO * 3] - px[HALO * 3]);
                if (r >= 2 * HALO) {
                    const int io = i0 + r - 2 * HALO;  // output row
                    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                    const bool interior = (io >= HALO) && (io < H - HALO) && (j >= HALO) && (j < W - HALO);
                    if (interior) {
                        float mx = 0, my = 0, exx = 0, eyy = 0, exy = 0;
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            const int slot = (s + 1 + k) % KS;  // input row r-10+k
                            const float w = win.w[k];
                            mx += w * ring[slot][0]; my += w * ring[slot][1]; exx += w * ring[slot][2];
                            eyy += w * ring[slot][3]; exy += w * ring[slot][4];
                        }
                        // variances clamped at 0 like torchmetrics (torch.clamp(E[x^2] - mu^2, min=0)): float cancellation on flat
                        // regions would otherwise leave them slightly negative; a clamped sigma_x^2 passes no gradient
                        const float sxx_raw = exx - mx * mx;
                        const float sxx = fmaxf(sxx_raw, 0.f), syy = fmaxf(eyy - my * my, 0.f), sxy = exy - mx * my;
                        const float n1 = 2.f * mx * my + c1, n2 = 2.f * sxy + c2;
                        const float dd1 = mx * mx + my * my + c1, dd2 = sxx + syy + c2;
                        // v_rcp_f32 (1 ulp) instead of three IEEE divisions (~10 instructions each): the kernel is
                        // VALU bound and the parity bar on SSIM is 1e-5
                        const float inv1 = __builtin_amdgcn_rcpf(dd1), inv2 = __builtin_amdgcn_rcpf(dd2);
                        const float inv = inv1 * inv2;
                        const float ssim = n1 * n2 * inv;
                        ssim_acc += ssim;
                        const float dn1 = n2 * inv, dn2 = n1 * inv;
                        const float g1 = -ssim * inv1, g2 = sxx_raw < 0.f ? 0.f : -ssim * inv2;
                        d0 = 2.f * my * (dn1 - dn2) + 2.f * mx * (g1 - g2);  // dS/dmu_x
                        d1 = g2;                                             // dS/dE[x^2]
                        d2 = 2.f * dn2;                                      // dS/dE[xy]
                    }
                    if (D && io < H && j < W) {
                        float* dst = D + ((((int64_t)cam * H + io) * W + j) * 3 + ch) * 3;
                        dst[0] = d0; dst[1] = d1; dst[2] = d2;
                    }
                }
                __syncthreads();
            }
        }
    }
    const float tl1 = block_sum_192(l1, red);
    const float tss = block_sum_192(ssim_acc, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * cam + 0], (double)tl1);
        atomicAdd(&sums[2 * cam + 1], (double)tss);
    }
}


// Round 5 (tools/probe/valu_issue.hip): a VALU instruction with an SGPR source takes a four-cycle slot of the CU's scalar
// operand port, one with VGPR sources only takes two cycles of its SIMD -- and three of every four multiply-adds of these
// kernels take a window weight.  The weights therefore live in VGPRs (the asm keeps hipcc from moving them back).
struct WinV { float w[KS]; };
__device__ __forceinline__ WinV window_in_vgprs(const Win& win) {
    WinV v;
#pragma unroll
    for (int k = 0; k <= HALO; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(v.w[k]) : "s"(win.w[k]));
#pragma unroll
    for (int k = HALO + 1; k < KS; ++k) v.w[k] = v.w[KS - 1 - k];   // (the window is symmetric: six registers)
    return v;
}

// ---- moments of the ground-truth image, once per training call (round 6) ----
// Two of the five windowed moments, conv(y) and conv(y^2), depend on the ground truth alone, and the ground truth does
// not change inside a run_3dgs_optim call (starster/gs.py:149-152 passes the same scene.imgs every iteration).
// st3r_loss_gt_moments computes them once, [C,H,W,3,2] floats = (conv(y), conv(y^2)) per (pixel, channel), zeros outside
// the interior; k_ssim_fused<true> reads them instead of convolving y: 3 instead of 5 moment maps in its forward waves
// (33 + 22 of ~275 VALU instructions per thread and row, 36 instead of 60 ring registers), 8 more bytes read per pixel and
// channel.  Both kernels take the y taps through the macros below -- one association, no contraction left to the
// compiler -- so the moments are the same bits whichever kernel computes them (tests/test_gpu_gs.py).
#define YMOM_FIRST(h1, h3, w, yy) { _Pragma("clang fp contract(off)") const float wy_ = (w) * (yy); h1 = wy_; h3 = wy_ * (yy); }
#define YMOM_TAP(h1, h3, w, yy) { _Pragma("clang fp contract(off)") const float wy_ = (w) * (yy); h1 = __builtin_fmaf((w), (yy), h1); h3 = __builtin_fmaf(wy_, (yy), h3); }
#define VMOM_FIRST(acc, w, v) { _Pragma("clang fp contract(off)") acc = (w) * (v); }
#define VMOM_TAP(acc, w, v) { acc = __builtin_fmaf((w), (v), acc); }

// One thread per (column, channel) of a 64-column strip, rows streamed through LDS, the last eleven horizontal results in
// a register ring -- k_ssim_fwd's structure with y alone.  Runs once per training call: not tuned.
__global__ __launch_bounds__(LT) void k_gt_moments(int LH, int H, int W, const float* __restrict__ gt, Win win_s,
                                                   float2* __restrict__ mom) {
    __shared__ float sy[2][SEG];
    const WinV win = window_in_vgprs(win_s);
    const int cam = blockIdx.z;
    const int j0 = blockIdx.x * LW, i0 = blockIdx.y * LH;
    const int t = threadIdx.x;
    const int col = t / 3, ch = t - col * 3;
    const int j = j0 + col;
    const float* yr = gt + (int64_t)cam * H * W * 3;
    const int nrows = min(LH, H - i0) + 2 * HALO;  // input rows i0-5 .. i0+rows+4
    constexpr int NPF = (SEG + LT - 1) / LT;
    auto stage = [&](int r, int b) {
        const int i = min(max(i0 - HALO + r, 0), H - 1);   // (clamped: a window of an interior pixel never leaves the image)
#pragma unroll
        for (int n = 0; n < NPF; ++n) {
            const int e = t + n * LT;
            if (e < SEG) {
                const int jj = min(max(j0 - HALO + e / 3, 0), W - 1);
                sy[b][e] = yr[((int64_t)i * W + jj) * 3 + (e - (e / 3) * 3)];
            }
        }
    };
    float ring[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) { ring[s][0] = 0.f; ring[s][1] = 0.f; }
    stage(0, 0);
    __syncthreads();
    for (int rb = 0; rb < nrows; rb += KS) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int r = rb + s;
            if (r < nrows) {
                const int b = r & 1;
                if (r + 1 < nrows) stage(r + 1, b ^ 1);
                const float* py = &sy[b][col * 3 + ch];
                float h1, h3;
                YMOM_FIRST(h1, h3, win.w[0], py[0])
#pragma unroll
                for (int k = 1; k < KS; ++k) YMOM_TAP(h1, h3, win.w[k], py[k * 3])
                ring[s][0] = h1; ring[s][1] = h3;
                if (r >= 2 * HALO) {
                    const int io = i0 + r - 2 * HALO;  // output row
                    float my = 0.f, eyy = 0.f;
                    if ((io >= HALO) && (io < H - HALO) && (j >= HALO) && (j < W - HALO)) {
                        VMOM_FIRST(my, win.w[0], ring[(s + 1) % KS][0])
                        VMOM_FIRST(eyy, win.w[0], ring[(s + 1) % KS][1])
#pragma unroll
                        for (int k = 1; k < KS; ++k) {
                            const int slot = (s + 1 + k) % KS;  // input row r-10+k
                            VMOM_TAP(my, win.w[k], ring[slot][0])
                            VMOM_TAP(eyy, win.w[k], ring[slot][1])
                        }
                    }
                    if (io < H && j < W) mom[(((int64_t)cam * H + io) * W + j) * 3 + ch] = make_float2(my, eyy);
                }
                __syncthreads();
            }
        }
    }
}

// Forward and backward in one pass: the derivative maps D never leave the CU.  The strip's output columns need D on
// 5 more columns each side, D needs x, y on 5 more again, so the workgroup stages 84-column row segments.
// Wave specialisation: waves 0-3 (222 active threads) run the forward for 74 columns (same arithmetic, same order as
// k_ssim_fwd) and leave each D row in LDS; waves 4-6 (192 threads) run the second convolution for the 64 output
// columns one row behind them -- the two stages overlap instead of alternating, each thread carries one 11-row
// register ring, one barrier per image row.  Saves the 9-float-per-pixel round trip through HBM (1.2 of the 2.6 GB
// the two-kernel version moves per iteration at 8 x 1080p).
#ifndef FLW
#define FLW 64                        // output columns of a fused strip
#endif
// Row loop of the fused kernel: unrolled by SSIM_U = 12 with 12-slot rings (one slot more than the 11 rows a window
// needs), so that both the ring slot of a row and the slot of the SSIM_PD-deep load queue (SSIM_PD divides 12) are
// compile-time constants; the loads of a row are unconditional (clamped addresses).  Round 4, measured: the row loop
// without any arithmetic takes 0.34 ms WITH its global loads and 0.12 ms without them -- but the whole kernel takes 0.51 ms
// without them and 0.52 with: the arithmetic hides the loads, one row of slack is enough (SSIM_PD = 1; deeper queues only
// cost registers -- and hipcc waits for every load in flight at the commit anyway: 0.59 / 0.60 ms at depth 2 / 3).
// What the restructuring did buy is the branch-free addressing: 0.58 -> 0.52 ms.
#ifndef SSIM_PD
#define SSIM_PD 1
#endif
#define SSIM_U 12
#define HALO2 (2 * HALO)
#define DCOLS (FLW + 2 * HALO)        // columns of D a workgroup computes (74)
#define FT1 ((DCOLS * 3 + 63) / 64 * 64)   // forward threads, whole waves (256)
#define FBW ((FLW * 3 + 63) / 64 * 64)     // backward threads, whole waves (192)
#define FT2 (FT1 + FBW)
#define SEG2 ((FLW + 2 * HALO2) * 3)  // floats per staged row segment of an image (252)
#define NPF2 ((SEG2 + FT1 - 1) / FT1) // staged elements per forward thread and row

// (five waves per SIMD = two workgroups of 7 waves per CU: at most 96 VGPRs; the forward threads' five 12-row rings are 60)
// GTM: conv(y), conv(y^2) come from `gtm` (st3r_loss_gt_moments) instead of being convolved here
#ifndef SSIM_GTM_WAVES
#define SSIM_GTM_WAVES 5
#endif
template <bool GTM>
__global__ __launch_bounds__(FT2) __attribute__((amdgpu_waves_per_eu(GTM ? SSIM_GTM_WAVES : 5))) void k_ssim_fused(int LH, int H, int W, const float* __restrict__ render,
                                                    const float* __restrict__ gt, const float2* __restrict__ gtm,
                                                    Win win_s, float k_l1, float k_ss,
                                                    double* __restrict__ sums, float* __restrict__ v_render) {
    __shared__ float sx[2][SEG2];
    __shared__ float sy[2][SEG2];
    __shared__ float sd[2][DCOLS * 9];
    __shared__ float red[2 * (FT1 / 64)];
    const WinV win = window_in_vgprs(win_s);
    const int cam = blockIdx.z;
    const int j0 = blockIdx.x * FLW, i0 = blockIdx.y * LH;
    const float* xr = render + (int64_t)cam * H * W * 3;
    const float* yr = gt + (int64_t)cam * H * W * 3;
    const int rows_out = min(LH, H - i0);
    const int nrows = rows_out + 2 * HALO2;  // input rows i0-10 .. i0+rows_out+9
    float l1 = 0.f, ssim_acc = 0.f;

    if (threadIdx.x < FT1) {
        // ================= forward waves: D column `col` = image column j0 - 5 + col =================
        const int t = threadIdx.x;
        const int col = t / 3, ch = t - col * 3;
        const bool active = t < DCOLS * 3;
        const int jd = j0 - HALO + col;
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        float qx[SSIM_PD][NPF2], qy[SSIM_PD][NPF2];   // rows r+1 .. r+SSIM_PD at the top of iteration r; row q in slot q % SSIM_PD
        // The loads are UNCONDITIONAL (row and column clamped into the image): a load behind a branch makes the compiler
        // wait for every load in flight at the next join, which undoes the queue.  What a clamped address delivers outside
        // the image is never used: windows of interior pixels lie inside the image, everything else is forced to zero.
        unsigned eoff[NPF2];   // element offset inside an image row (column clamped), constant over the rows
#pragma unroll
        for (int n = 0; n < NPF2; ++n) {
            const int e = min(t + n * FT1, SEG2 - 1);
            const int jj = min(max(j0 - HALO2 + e / 3, 0), W - 1);
            eoff[n] = (unsigned)(jj * 3 + (e - (e / 3) * 3));
        }
        auto fetch_row = [&](int r, float (&vxs)[NPF2], float (&vys)[NPF2]) {
            // (the row is uniform over the workgroup: a scalar base pointer per row + one 32-bit offset per thread -- no
            // 64-bit address arithmetic on the vector unit)
            const int i = min(max(i0 - HALO2 + r, 0), H - 1);
            const float* xrow = xr + (int64_t)i * (W * 3);
            const float* yrow = yr + (int64_t)i * (W * 3);
#pragma unroll
            for (int n = 0; n < NPF2; ++n) { vxs[n] = xrow[eoff[n]]; vys[n] = yrow[eoff[n]]; }
        };
        auto commit_from = [&](int b, const float (&vxs)[NPF2], const float (&vys)[NPF2]) {
#pragma unroll
            for (int n = 0; n < NPF2; ++n) {
                const int e = t + n * FT1;
                if (e < SEG2) { sx[b][e] = vxs[n]; sy[b][e] = vys[n]; }
            }
        };
        constexpr int NM = GTM ? 3 : 5;   // ring entries per row: x, x^2, xy (and y, y^2)
        float ring[SSIM_U][NM];
#pragma unroll
        for (int s = 0; s < SSIM_U; ++s)
#pragma unroll
            for (int m = 0; m < NM; ++m) ring[s][m] = 0.f;
        const bool own_col = (col >= HALO) && (col < HALO + FLW) && (jd < W);
        // GTM: (conv(y), conv(y^2)) of the D row a later iteration completes, requested one iteration ahead like the image
        // rows (clamped, unconditional: see below); iteration r completes image row i0 + r - 15
        const float2* gcol = gtm + ((int64_t)cam * H * W + min(max(jd, 0), W - 1)) * 3 + ch;
        auto fetch_gtm = [&](int r) -> float2 {
            const int i = min(max(i0 + r - 3 * HALO, 0), H - 1);
            return gcol[(int64_t)i * (W * 3)];
        };
        float2 gq = make_float2(0.f, 0.f);
        if (GTM) gq = fetch_gtm(0);
        fetch_row(0, qx[0], qy[0]);
        commit_from(0, qx[0], qy[0]);
#pragma unroll
        for (int d = 1; d <= SSIM_PD; ++d) fetch_row(d, qx[d % SSIM_PD], qy[d % SSIM_PD]);   // rows 1 .. SSIM_PD are under way
        __syncthreads();
        for (int rb = 0; rb <= nrows; rb += SSIM_U) {
#pragma unroll
            for (int s = 0; s < SSIM_U; ++s) {
                const int r = rb + s;
                if (r <= nrows) {
                    if (r < nrows) {
                        const int b = r & 1;
                        commit_from(b ^ 1, qx[(s + 1) % SSIM_PD], qy[(s + 1) % SSIM_PD]);   // row r+1, requested SSIM_PD iterations ago
                        fetch_row(r + 1 + SSIM_PD, qx[(s + 1) % SSIM_PD], qy[(s + 1) % SSIM_PD]);
                        const float2 gcur = gq;
                        if (GTM) gq = fetch_gtm(r + 1);
                        if (active) {
                            const float* px = &sx[b][col * 3 + ch];
                            const float* py = &sy[b][col * 3 + ch];
                            // (the first tap initialises the sums: "0 + w x" would cost a move and a multiply-add where a
                            // multiply does -- sixteen instructions per image row over the four convolutions of this kernel)
                            float h0, h1 = 0.f, h2, h3 = 0.f, h4;
                            {
                                const float xx = px[0], yy = py[0], w = win.w[0];
                                h0 = w * xx; h2 = h0 * xx; h4 = h0 * yy;
                                if (!GTM) YMOM_FIRST(h1, h3, w, yy)
                            }
#pragma unroll
                            for (int k = 1; k < KS; ++k) {
                                const float xx = px[k * 3], yy = py[k * 3], w = win.w[k];
                                const float wx = w * xx;
                                h0 += wx; h2 += wx * xx; h4 += wx * yy;
                                if (!GTM) YMOM_TAP(h1, h3, w, yy)
                            }
                            ring[s][0] = h0; ring[s][1] = h2; ring[s][2] = h4;
                            if (!GTM) { ring[s][NM - 2] = h1; ring[s][NM - 1] = h3; }
                            const int i = i0 - HALO2 + r;  // image row just staged
                            if (own_col && i >= i0 && i < i0 + rows_out) l1 += fabsf(py[HALO * 3] - px[HALO * 3]);
                            if (r >= 2 * HALO) {
                                const int id = i0 + r - 3 * HALO;  // image row of the D row completed now
                                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                                const bool interior = (id >= HALO) && (id < H - HALO) && (jd >= HALO) && (jd < W - HALO);
                                if (interior) {
                                    float mx, my = gcur.x, exx, eyy = gcur.y, exy;
                                    {
                                        const int slot = (s + SSIM_U - 10) % SSIM_U;
                                        const float w = win.w[0];
                                        mx = w * ring[slot][0]; exx = w * ring[slot][1]; exy = w * ring[slot][2];
                                        if (!GTM) { VMOM_FIRST(my, w, ring[slot][NM - 2]) VMOM_FIRST(eyy, w, ring[slot][NM - 1]) }
                                    }
#pragma unroll
                                    for (int k = 1; k < KS; ++k) {
                                        const int slot = (s + SSIM_U - 10 + k) % SSIM_U;  // input row r-10+k
                                        const float w = win.w[k];
                                        mx += w * ring[slot][0]; exx += w * ring[slot][1]; exy += w * ring[slot][2];
                                        if (!GTM) { VMOM_TAP(my, w, ring[slot][NM - 2]) VMOM_TAP(eyy, w, ring[slot][NM - 1]) }
                                    }
                                    // variances clamped at 0 like torchmetrics (torch.clamp(E[x^2] - mu^2, min=0)): float cancellation on flat
                        // regions would otherwise leave them slightly negative; a clamped sigma_x^2 passes no gradient
                        const float sxx_raw = exx - mx * mx;
                        const float sxx = fmaxf(sxx_raw, 0.f), syy = fmaxf(eyy - my * my, 0.f), sxy = exy - mx * my;
                                    const float n1 = 2.f * mx * my + c1, n2 = 2.f * sxy + c2;
                                    const float dd1 = mx * mx + my * my + c1, dd2 = sxx + syy + c2;
                                    const float inv1 = __builtin_amdgcn_rcpf(dd1), inv2 = __builtin_amdgcn_rcpf(dd2);
                                    const float inv = inv1 * inv2;
                                    const float ssim = n1 * n2 * inv;
                                    if (own_col && id >= i0 && id < i0 + rows_out) ssim_acc += ssim;  // one strip counts it
                                    const float dn1 = n2 * inv, dn2 = n1 * inv;
                                    const float g1 = -ssim * inv1, g2 = sxx_raw < 0.f ? 0.f : -ssim * inv2;
                                    d0 = 2.f * my * (dn1 - dn2) + 2.f * mx * (g1 - g2);  // dS/dmu_x
                                    d1 = g2;                                             // dS/dE[x^2]
                                    d2 = 2.f * dn2;                                      // dS/dE[xy]
                                }
                                float* dst = &sd[b][col * 9 + ch * 3];
                                dst[0] = d0; dst[1] = d1; dst[2] = d2;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    } else {
        // ================= backward waves: output column j0 + col, one row behind the forward waves =================
        const int t = threadIdx.x - FT1;
        const int col = t / 3, ch = t - col * 3;
        const int j = t < FLW * 3 ? j0 + col : W;   // threads past the strip (whole-wave padding) only keep the barriers
        float ring[SSIM_U][3];
#pragma unroll
        for (int s = 0; s < SSIM_U; ++s) { ring[s][0] = 0.f; ring[s][1] = 0.f; ring[s][2] = 0.f; }
        // x, y of this thread's output pixel: iteration r writes image row i0 + r - 21; requested SSIM_PD iterations ahead
        // (slot r % SSIM_PD), like the forward waves' rows
        float bx[SSIM_PD], by[SSIM_PD];
        const unsigned jc = (unsigned)(min(j, W - 1) * 3 + ch);   // (unconditional, clamped loads: see the forward waves)
        auto fetch_px = [&](int r_use, float& vx, float& vy) {
            const int ion = min(max(i0 + r_use - 1 - 2 * HALO2, 0), H - 1);
            vx = (xr + (int64_t)ion * (W * 3))[jc]; vy = (yr + (int64_t)ion * (W * 3))[jc];
        };
#pragma unroll
        for (int d = 0; d < SSIM_PD; ++d) fetch_px(d, bx[d], by[d]);
        __syncthreads();
        for (int rb = 0; rb <= nrows; rb += SSIM_U) {
#pragma unroll
            for (int s = 0; s < SSIM_U; ++s) {
                const int r = rb + s;
                if (r <= nrows) {
                    const int rp = r - 1;                 // the forward iteration whose D row is consumed now
                    const int sp = (s + SSIM_U - 1) % SSIM_U;     // its ring slot (compile time)
                    const float x = bx[s % SSIM_PD], y = by[s % SSIM_PD];
                    fetch_px(r + SSIM_PD, bx[s % SSIM_PD], by[s % SSIM_PD]);
                    if (rp >= 2 * HALO && j < W) {
                        const float* pd = &sd[rp & 1][col * 9 + ch * 3];
                        float h0 = win.w[0] * pd[0], h1 = win.w[0] * pd[1], h2 = win.w[0] * pd[2];
#pragma unroll
                        for (int k = 1; k < KS; ++k) {
                            const float w = win.w[k];
                            h0 += w * pd[k * 9]; h1 += w * pd[k * 9 + 1]; h2 += w * pd[k * 9 + 2];
                        }
                        ring[sp][0] = h0; ring[sp][1] = h1; ring[sp][2] = h2;
                        if (rp >= 2 * HALO2) {
                            const int io = i0 + rp - 2 * HALO2;
                            if (io < H && j < W) {
                                float a0, a1, a2;
                                {
                                    const int slot = (sp + SSIM_U - 10) % SSIM_U;
                                    a0 = win.w[0] * ring[slot][0]; a1 = win.w[0] * ring[slot][1]; a2 = win.w[0] * ring[slot][2];
                                }
#pragma unroll
                                for (int k = 1; k < KS; ++k) {
                                    const int slot = (sp + SSIM_U - 10 + k) % SSIM_U;
                                    const float w = win.w[k];
                                    a0 += w * ring[slot][0]; a1 += w * ring[slot][1]; a2 += w * ring[slot][2];
                                }
                                float* vrow = v_render + ((int64_t)cam * H + io) * (W * 3);   // (uniform row pointer)
                                const float sgn = (x > y) ? 1.0f : ((x < y) ? -1.0f : 0.0f);
                                vrow[(unsigned)(j * 3 + ch)] = k_l1 * sgn + k_ss * (a0 + 2.f * x * a1 + y * a2);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
    // sums of the forward waves (the others hold zeros)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_down(l1, off); ssim_acc += __shfl_down(ssim_acc, off); }
    constexpr int FW = FT1 / 64;
    if ((threadIdx.x & 63) == 0 && threadIdx.x < FT1) { red[threadIdx.x >> 6] = l1; red[FW + (threadIdx.x >> 6)] = ssim_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < FW; ++i) { a += red[i]; b += red[FW + i]; }
        atomicAdd(&sums[2 * cam + 0], (double)a);
        atomicAdd(&sums[2 * cam + 1], (double)b);
    }
}

// the registered ground-truth moments that belong to the images at `gt` (NULL: none -- the kernel convolves y itself)
static const float2* gt_moments_for(const st3r_ctx* ctx, const float* gt, int C, int H, int W) {
    if (!ctx->gtm_mom || H != ctx->gtm_h || W != ctx->gtm_w) return nullptr;
    const int64_t img = (int64_t)H * W * 3;
    if (gt < ctx->gtm_gt || (gt - ctx->gtm_gt) % img != 0) return nullptr;
    const int64_t c0 = (gt - ctx->gtm_gt) / img;
    if (c0 + C > ctx->gtm_c) return nullptr;
    return reinterpret_cast<const float2*>(ctx->gtm_mom) + c0 * img;
}

int st3r_loss_impl(st3r_ctx* ctx, hipStream_t s, int C, int H, int W, const float* render, const float* gt,
                   float w_l1, float w_ssim, double* sums, float* v_render, bool sums_cleared) {
    static const Win win = make_window();
    if (!sums_cleared) HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)C, s));
    // A strip of LH output rows streams LH + 20 input rows: tall strips waste less, but the launch needs a few
    // workgroups per CU -- as few row strips as still give ~1000 workgroups, at least 64 rows each
#ifndef SSIM_TARGET_WGS
#define SSIM_TARGET_WGS 1020
#endif
    const int strip = v_render ? FLW : LW;
    const int per_band = ceil_div(W, strip) * C;
    // (measured, tools/experiments/ssim_bands.sh: ~1000 workgroups, strips of at least 64 rows -- 8 views: 5 bands, 1 view: 17)
    int bands = std::max(1, std::min(ceil_div(SSIM_TARGET_WGS, per_band), std::max(1, H / 64)));
    const float2* gtm = v_render ? gt_moments_for(ctx, gt, C, H, W) : nullptr;
    // k_ssim_fused<true> (80 VGPRs) keeps THREE workgroups per CU: as many row bands as still fit the 768 resident slots in
    // one round (round 6, 8 x 1080p: 3 bands = 720 workgroups 0.404 ms, 6 bands 0.403, 4 bands 0.411, the 5 bands of the
    // rule above 0.453, 8 bands 0.437, 10 bands 0.416: what matters is how the workgroups fill whole rounds of the chip)
    // (the same for the kernel that convolves the ground truth itself -- 94 VGPRs, two workgroups per CU, 512 slots: 8 x 1080p
    // with 2 bands 0.436 ms, 3 bands 0.49, 4 bands 0.45, 5 bands 0.47, 6 bands 0.46)
    if (v_render) bands = std::max(1, std::min((gtm ? 768 : 512) / std::max(per_band, 1), std::max(1, H / 64)));
    if (const char* e = getenv("ST3R_SSIM_BANDS")) bands = std::max(1, atoi(e));   // tuning hook (tools/experiments/ssim_bands.sh)
    const int LH = ceil_div(H, bands);
    dim3 grid(ceil_div(W, strip), ceil_div(H, LH), C);
    if (!v_render) {   // loss value only
        hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(LT), 0, s, LH, H, W, render, gt, win, sums, (float*)nullptr);
        LAUNCH_CHECK();
        return ST3R_OK;
    }
    const int Hi = H - 2 * HALO, Wi = W - 2 * HALO;
    const double cnt = (Hi > 0 && Wi > 0) ? (double)Hi * Wi * 3 : 0.0;
    const float k_l1 = (float)((double)w_l1 / ((double)H * W * 3));
    const float k_ss = cnt > 0 ? (float)(-(double)w_ssim / cnt) : 0.f;
    if (gtm)
        hipLaunchKernelGGL(k_ssim_fused<true>, grid, dim3(FT2), 0, s, LH, H, W, render, gt, gtm, win, k_l1, k_ss, sums, v_render);
    else
        hipLaunchKernelGGL(k_ssim_fused<false>, grid, dim3(FT2), 0, s, LH, H, W, render, gt, gtm, win, k_l1, k_ss, sums, v_render);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_loss_gt_moments(st3r_ctx* ctx, void* stream, int C, int height, int width, const float* gt,
                                     float* moments) {
    ARG_CHECK(ctx && C > 0 && height > 0 && width > 0 && gt && moments);
    static const Win win = make_window();
    const int H = height, W = width;
    const int bands = std::max(1, std::min(ceil_div(1020, ceil_div(W, LW) * C), std::max(1, H / 64)));
    const int LH = ceil_div(H, bands);
    hipLaunchKernelGGL(k_gt_moments, dim3(ceil_div(W, LW), ceil_div(H, LH), C), dim3(LT), 0, (hipStream_t)stream, LH, H, W, gt,
                       win, reinterpret_cast<float2*>(moments));
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_ctx_set_gt_moments(st3r_ctx* ctx, const float* gt, const float* moments, int C, int height,
                                        int width) {
    ARG_CHECK(ctx);
    if (!gt || !moments) { ctx->gtm_gt = nullptr; ctx->gtm_mom = nullptr; ctx->gtm_c = ctx->gtm_h = ctx->gtm_w = 0; return ST3R_OK; }
    ARG_CHECK(C > 0 && height > 0 && width > 0);
    ctx->gtm_gt = gt; ctx->gtm_mom = moments; ctx->gtm_c = C; ctx->gtm_h = height; ctx->gtm_w = width;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_loss_l1_ssim(st3r_ctx* ctx, void* stream, int C, int height, int width, const float* render,
                                  const float* gt, float w_l1, float w_ssim, double* sums, float* v_render) {
    ARG_CHECK(ctx && C > 0 && height > 0 && width > 0 && render && gt && sums);
    return st3r_loss_impl(ctx, (hipStream_t)stream, C, height, width, render, gt, w_l1, w_ssim, sums, v_render, false);
}
