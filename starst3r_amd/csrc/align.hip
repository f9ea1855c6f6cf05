// K11: global alignment (path B) -- fused residual + analytic gradient + Adam.
// Replaces the reference's optimisation loop starster/reconstruct.py:371-406 (`optimize_loop`,
// called for the coarse 3-D stage :427 and the 2-D reprojection stage :440) together with
// `make_K_cam_depth` (:209-261), `loss_3d` (:325-353), `loss_2d` (:355-369), `loss_dust3r` (:311-323)
// and the make_pts3d / reproj2d / gamma_loss helpers it imports.
//
// The reference builds a ~6400-op autograd graph per iteration (launch/dispatch bound).
// Here one iteration is two launches, with hand-derived gradients and no host round trip:
//   k_align_resid  : one thread per correspondence row.  Rebuilds the two 3-D points from the anchors,
//                    evaluates conf * rho(distance) and pushes the gradient back to the 17 quantities of
//                    each touched camera (R[9] T[3] f cx cy A B, depth = A + B*core).  Per-workgroup LDS
//                    accumulators (ds_add_f32), then one float atomic per slot per workgroup.
//   k_align_update : ONE workgroup.  Back-propagates the per-camera sums through the reparametrised
//                    translation, the MST kinematic chain, the focal clamp and global_scaling = 1/min(size)
//                    to the 11 parameters per view, applies Adam(lr(t), betas (0.9,0.9), eps 1e-8) with the
//                    cosine schedule, renormalises the quaternions (:394-395), and immediately rebuilds the
//                    camera table for the next iteration.
// The working set (a few MB) lives in L2/LDS: this path is latency bound, not roofline bound.
//
// Parameter conventions of the reference: quaternions are (x,y,z,w) (:150), principal
// points are normalised by the image size (:170), core depths by their median (:176-177).
#include <vector>

#include "common.h"

#define CAM_STRIDE 24   // R[9] T[3] f cx cy A B base_focal (+pad)
#define ACC_STRIDE 20   // vR[9] vT[3] vf vcx vcy vA vB (+pad)
#define MAXC 256

struct AlignProblem {
    int C;
    const float* imsizes;      // [C,2] (W,H) as float
    const float* base_focals;  // [C]
    const float* median;       // [C]   median of the raw core depths
    const float* core;         // [C,G] core depths / median
    int G;
    const float* anchor_pix;   // [A,2]
    const int32_t* anchor_idx; // [A]
    const float* anchor_off;   // [A]
    const int32_t* anchor_img; // [A]
    int n_corr; const int32_t* corr_a1; const int32_t* corr_a2; const float* corr_w;   // w = conf / sum(conf)
    int n_c2d; const float* c2d_pix; const int32_t* c2d_a2; const int32_t* c2d_img1; const float* c2d_w;
    int n_dust; const int32_t* dust_a1; const float* dust_tgt; const int32_t* dust_img2; const float* dust_w;
    int root; int n_edges; const int32_t* edges;  // [n_edges,2] in chain order
    const float* min_focals; const float* max_focals;
    const float4* anchor_pack;  // [A,2] packed once per run: (u, v, core value, offset) | (img as int bits, 0, 0, 0)
    // robust losses gamma_loss(gamma) of the three row kinds (reconstruct.py:118-120: loss1, loss2, lossd) and their
    // offsets (1/gamma)^(1/(gamma-1)); gamma = 1 is the plain L1 distance (offset 0)
    float gamma1, off1, gamma2, off2, gammad, offd;
    // opt_depth (reconstruct.py:437): stage 2 also leaves, per row, dL/d(core depth of the row's anchor); NULL otherwise
    float* rowgrad;
};

struct AlignState {
    float* pps; float* log_focals; float* quats; float* trans; float* log_sizes;  // [C,2] [C] [C,4] [C,3] [C]
    float* m; float* v;        // Adam moments [11*C] in the order pps, log_focals, quats, trans, log_sizes
    float* cam;                // [C,CAM_STRIDE]
    float* acc;                // [C*ACC_STRIDE + 4]: gradient sums, then [loss, nan flag]
    float* part;               // [workgroups of k_align_resid][C*ACC_STRIDE + 1]: their partial sums
    float* losses;             // [niter1 + niter2]
    float* chain;              // [12*C]: chained rotations [9C] and translations [3C] of the CURRENT parameters, left by the
                               // forward half of k_align_update for the backward half of the next call (or NULL)
};

__device__ __forceinline__ float rho_prime(float d, float gamma, float off, float* rho) {
    // gamma_loss(gamma): rho(d) = (d + off)^gamma - off^gamma, off = (1/gamma)^(1/(gamma-1))
    if (gamma == 1.0f) { *rho = d; return 1.0f; }
    const float b = d + off;
    const float pw = __powf(b, gamma - 1.0f);
    *rho = pw * b - __powf(off, gamma);
    return gamma * pw;
}

// 3-D point of anchor a in world coordinates; also returns the pieces the backward pass needs
struct Pt { float pw[3]; float pc[3]; float dx, dy, D, offp, core, off, inv_f; int img; };

__device__ __forceinline__ Pt anchor_point(const AlignProblem& P, const float* __restrict__ cam, int a) {
    Pt r;
    // one dependent load level: everything constant about the anchor was packed up front (k_align_pack_anchors)
    const float4 p0 = P.anchor_pack[2 * a], p1 = P.anchor_pack[2 * a + 1];
    r.img = __float_as_int(p1.x);
    const float* c = cam + r.img * CAM_STRIDE;
    const float f = c[12], cx = c[13], cy = c[14], A = c[15], B = c[16], bf = c[17];
    const float u = p0.x, v = p0.y;
    r.core = p0.z;
    r.D = A + B * r.core;
    r.off = p0.w;
    r.inv_f = 1.0f / f;   // one division per point; the kernel runs one wave per SIMD, so instruction count is time
    r.offp = 1.0f + (p0.w - 1.0f) * (bf * r.inv_f);
    const float z = r.D * r.offp;
    r.dx = (u - cx) * r.inv_f; r.dy = (v - cy) * r.inv_f;
    r.pc[0] = z * r.dx; r.pc[1] = z * r.dy; r.pc[2] = z;
    r.pw[0] = c[0] * r.pc[0] + c[1] * r.pc[1] + c[2] * r.pc[2] + c[9];
    r.pw[1] = c[3] * r.pc[0] + c[4] * r.pc[1] + c[5] * r.pc[2] + c[10];
    r.pw[2] = c[6] * r.pc[0] + c[7] * r.pc[1] + c[8] * r.pc[2] + c[11];
    return r;
}

__global__ void k_align_pack_anchors(int n, const float* __restrict__ pix, const int32_t* __restrict__ idx,
                                     const float* __restrict__ off, const int32_t* __restrict__ img,
                                     const float* __restrict__ core, int G, float4* __restrict__ pack) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int im = img[a];
    pack[2 * a] = make_float4(pix[2 * a], pix[2 * a + 1], core[(int64_t)im * G + idx[a]], off[a]);
    pack[2 * a + 1] = make_float4(__int_as_float(im), 0.f, 0.f, 0.f);
}

// One row contributes to at most two cameras: 17 numbers each (vR[9] vT[3] vf vcx vcy vA vB), collected in
// registers and handed to the LDS accumulators once, outside all divergent control flow (flush_camera).
struct CamGrad { int img; float g[17]; };

__device__ __forceinline__ void cam_clear(CamGrad& c) {
    c.img = -1;
#pragma unroll
    for (int k = 0; k < 17; ++k) c.g[k] = 0.f;
}

// dL/d(pw) of an anchor point -> its camera's 17 numbers
__device__ __forceinline__ float point_bwd(const AlignProblem& P, const float* __restrict__ cam, int a, const Pt& r,
                                           const float vp[3], CamGrad& out) {
    const float* c = cam + r.img * CAM_STRIDE;
    float* g = out.g;
    out.img = r.img;
    const float bf = c[17];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i * 3 + j] += vp[i] * r.pc[j];
        g[9 + i] += vp[i];
    }
    const float vpc0 = c[0] * vp[0] + c[3] * vp[1] + c[6] * vp[2];
    const float vpc1 = c[1] * vp[0] + c[4] * vp[1] + c[7] * vp[2];
    const float vpc2 = c[2] * vp[0] + c[5] * vp[1] + c[8] * vp[2];
    const float z = r.pc[2];
    const float vz = vpc0 * r.dx + vpc1 * r.dy + vpc2;
    const float vdx = vpc0 * z, vdy = vpc1 * z;
    float vf = -(vdx * r.dx + vdy * r.dy) * r.inv_f;
    g[13] += -vdx * r.inv_f;
    g[14] += -vdy * r.inv_f;
    const float vD = vz * r.offp;
    g[15] += vD;
    g[16] += vD * r.core;
    vf += vz * r.D * (-(r.off - 1.0f) * bf * (r.inv_f * r.inv_f));
    g[12] += vf;
    return vD * c[16];   // dL/d(core depth): depth = A + B core
}

// Rows arrive grouped by image pair, so a wave's 64 rows almost always feed the same camera: 64 lanes adding to the
// same 17 LDS words serialise badly (it was ~2/3 of the kernel).  When the wave agrees on the camera the 17 numbers
// are summed across the wave with shuffles and one lane adds them; otherwise every lane adds its own.
// Sum over the 64 lanes with DPP only (four steps inside the 16-lane rows, then row_bcast:15 / row_bcast:31 chain
// the rows): the total ends up in every lane of row 3 (lanes 48..63).  All lanes must be active.  __shfl_xor would
// go through ds_bpermute -- an LDS round trip per step, ~60 ns each in this one-wave-per-SIMD kernel.
__device__ __forceinline__ float wave_sum_row3(float v) {
    asm volatile(
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "+v"(v));
    return v;
}

// `sw` = THIS WAVE's accumulator array: nothing else writes it, so plain read-modify-writes by one lane per word are
// enough and the order of the additions is the program order -- gradients are bit-reproducible run to run (round 1
// added into one array shared by the four waves with LDS float atomics, whose order is not defined).
__device__ __forceinline__ void flush_camera(const CamGrad& c, float* sw) {
    uint64_t todo = __ballot(c.img >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {   // one trip per distinct camera among the wave's rows (almost always exactly one), in lane order
        const int ref = __builtin_amdgcn_readlane(c.img, __builtin_ctzll(todo));
        const bool mine = c.img == ref;
        float* g = sw + ref * ACC_STRIDE;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const float v = wave_sum_row3(mine ? c.g[k] : 0.f);   // all lanes active here: todo is wave-uniform
            if (lane == 48 + (k & 15)) g[k] += v;                  // row 3 holds the totals: 17 distinct words
        }
        todo &= ~__ballot(mine);
    }
}

// stage: 1 = loss_3d rows + dust rows, 2 = loss_2d rows + dust rows
// (`wg`: index of the workgroup among those of the residual phase)
__device__ __forceinline__ void align_resid_body(const AlignProblem& P, const AlignState& S, int stage, float dust_w, int rpt,
                                                 int wg) {
    extern __shared__ float sacc[];  // 4 x [C*ACC_STRIDE + 1] accumulators (one per wave), then the camera table [C*CAM_STRIDE]
    const int nacc = P.C * ACC_STRIDE + 1;
    float* scam = sacc + 4 * nacc;
    float* sw = sacc + (threadIdx.x >> 6) * nacc;
    for (int i = threadIdx.x; i < 4 * nacc; i += blockDim.x) sacc[i] = 0.f;
    for (int i = threadIdx.x; i < P.C * CAM_STRIDE; i += blockDim.x) scam[i] = S.cam[i];   // overlaps the row index loads
    __syncthreads();
    const bool running = *(S.acc + P.C * ACC_STRIDE + 1) == 0.f;   // not stopped by a NaN loss
    for (int rr = 0; rr < rpt; ++rr) {   // rpt consecutive groups of 256 rows per workgroup (bounds the number of partials)
    CamGrad ca, cb;
    cam_clear(ca); cam_clear(cb);
    float lsum = 0.f, vcore = 0.f;
    if (running) {
        const int n_main = stage == 1 ? P.n_corr : P.n_c2d;
        const int row = (wg * rpt + rr) * blockDim.x + threadIdx.x;
        if (row < n_main) {
            if (stage == 1) {
                const int a1 = P.corr_a1[row], a2 = P.corr_a2[row];
                const Pt p1 = anchor_point(P, scam, a1), p2 = anchor_point(P, scam, a2);
                const float ex = p1.pw[0] - p2.pw[0], ey = p1.pw[1] - p2.pw[1], ez = p1.pw[2] - p2.pw[2];
                const float d = sqrtf(ex * ex + ey * ey + ez * ez);
                float rho;
                const float rp = rho_prime(d, P.gamma1, P.off1, &rho);
                const float w = P.corr_w[row];
                lsum = w * rho;
                if (d > 1e-20f) {
                    const float k = w * rp / d;
                    const float v1[3] = {k * ex, k * ey, k * ez}, v2[3] = {-k * ex, -k * ey, -k * ez};
                    point_bwd(P, scam, a1, p1, v1, ca);
                    point_bwd(P, scam, a2, p2, v2, cb);
                }
            } else {
                const int a2 = P.c2d_a2[row], i1 = P.c2d_img1[row];
                const Pt p2 = anchor_point(P, scam, a2);
                const float* c = scam + i1 * CAM_STRIDE;
                const float f = c[12], cx = c[13], cy = c[14];
                const float e0 = p2.pw[0] - c[9], e1 = p2.pw[1] - c[10], e2 = p2.pw[2] - c[11];
                const float qx = c[0] * e0 + c[3] * e1 + c[6] * e2;   // R1^T (p - T1)
                const float qy = c[1] * e0 + c[4] * e1 + c[7] * e2;
                const float qz = c[2] * e0 + c[5] * e1 + c[8] * e2;
                const float rx = f * qx + cx * qz, ry = f * qy + cy * qz, rz = qz;
                const bool zclip = !(rz >= 1e-3f);
                const float zc = zclip ? 1e-3f : rz;
                const float inv_zc = 1.0f / zc;
                float u = rx * inv_zc, v = ry * inv_zc;
                const bool uclip = (u < -1000.f) || (u > 2000.f), vclip = (v < -1000.f) || (v > 2000.f);
                u = fminf(fmaxf(u, -1000.f), 2000.f); v = fminf(fmaxf(v, -1000.f), 2000.f);
                const float du = P.c2d_pix[2 * row] - u, dv = P.c2d_pix[2 * row + 1] - v;
                const float d = sqrtf(du * du + dv * dv);
                float rho;
                const float rp = rho_prime(d, P.gamma2, P.off2, &rho);
                const float w = P.c2d_w[row];
                lsum = w * rho;
                if (d > 1e-20f) {
                    const float k = w * rp / d;
                    const float vu = uclip ? 0.f : -k * du, vv = vclip ? 0.f : -k * dv;
                    const float vrx = vu * inv_zc, vry = vv * inv_zc;
                    const float vrz = zclip ? 0.f : -(vu * rx + vv * ry) * (inv_zc * inv_zc);
                    float* g = ca.g;
                    ca.img = i1;
                    g[12] += vrx * qx + vry * qy;
                    g[13] += vrx * qz;
                    g[14] += vry * qz;
                    const float vq0 = f * vrx, vq1 = f * vry, vq2 = cx * vrx + cy * vry + vrz;
                    // q = R1^T e : vR1[m][k] += e_m vq_k ; ve = R1 vq ; vT1 -= ve ; vp += ve
                    const float e[3] = {e0, e1, e2}, vq[3] = {vq0, vq1, vq2};
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk) g[m * 3 + kk] += e[m] * vq[kk];
                    float ve[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) ve[m] = c[m * 3] * vq0 + c[m * 3 + 1] * vq1 + c[m * 3 + 2] * vq2;
#pragma unroll
                    for (int m = 0; m < 3; ++m) g[9 + m] += -ve[m];
                    vcore = point_bwd(P, scam, a2, p2, ve, cb);
                }
            }
        } else if (row - n_main < P.n_dust) {
            // DUSt3R regression fallback for pairs that failed the matching gate (reconstruct.py:311-323)
            const int r = row - n_main;
            const int a1 = P.dust_a1[r], i2 = P.dust_img2[r];
            const Pt p1 = anchor_point(P, scam, a1);
            const float* c = scam + i2 * CAM_STRIDE;
            const float t0 = P.dust_tgt[3 * r], t1 = P.dust_tgt[3 * r + 1], t2 = P.dust_tgt[3 * r + 2];
            const float gx = c[0] * t0 + c[1] * t1 + c[2] * t2 + c[9];
            const float gy = c[3] * t0 + c[4] * t1 + c[5] * t2 + c[10];
            const float gz = c[6] * t0 + c[7] * t1 + c[8] * t2 + c[11];
            const float ex = p1.pw[0] - gx, ey = p1.pw[1] - gy, ez = p1.pw[2] - gz;
            const float d = sqrtf(ex * ex + ey * ey + ez * ez);
            float rho;
            const float rp = rho_prime(d, P.gammad, P.offd, &rho);
            const float w = dust_w * P.dust_w[r];
            lsum = w * rho;
            if (d > 1e-20f) {
                const float k = w * rp / d;
                const float v1[3] = {k * ex, k * ey, k * ez};
                vcore = point_bwd(P, scam, a1, p1, v1, ca);
                float* g = cb.g;
                cb.img = i2;
                const float tg[3] = {t0, t1, t2};
#pragma unroll
                for (int m = 0; m < 3; ++m) {
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) g[m * 3 + kk] += -v1[m] * tg[kk];
                    g[9 + m] += -v1[m];
                }
            }
        }
    }
    if (P.rowgrad && stage == 2) {   // opt_depth: one number per row, summed per core depth by k_align_depth_update
        const int row = (wg * rpt + rr) * blockDim.x + threadIdx.x;
        if (row < P.n_c2d + P.n_dust) P.rowgrad[row] = vcore;
    }
    // converged again: wave-level hand-over to the wave's LDS accumulators
    flush_camera(ca, sw);
    flush_camera(cb, sw);
    {
        const float v = wave_sum_row3(lsum);
        if ((threadIdx.x & 63) == 63) sw[P.C * ACC_STRIDE] += v;
    }
    }   // rr
    __syncthreads();
    // the workgroup's partial sums (four waves in a fixed order); k_align_update adds the workgroups in order: no float
    // atomics anywhere, so the whole alignment is bit-reproducible
    float* part = S.part + (size_t)wg * nacc;
    for (int i = threadIdx.x; i < nacc; i += blockDim.x)
        part[i] = (sacc[i] + sacc[nacc + i]) + (sacc[2 * nacc + i] + sacc[3 * nacc + i]);
}

__global__ __launch_bounds__(256) void k_align_resid(AlignProblem P, AlignState S, int stage, float dust_w, int rpt) {
    align_resid_body(P, S, stage, dust_w, rpt, blockIdx.x);
}

__device__ __forceinline__ void quat_to_rot(const float* q, float* R, float* qn, float* inv_norm) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
    qn[0] = x; qn[1] = y; qn[2] = z; qn[3] = w; *inv_norm = inv;
}

// S.acc[k] = sum over the residual workgroups of their partials, in workgroup order (used when there are too many of
// them for the single workgroup of k_align_update to add without stretching the iteration)
__global__ __launch_bounds__(256) void k_align_reduce(AlignState S, int nacc, int n_part) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nacc) return;
    const float* __restrict__ part = S.part;
    float a = 0.f;
    for (int b0 = 0; b0 < n_part; b0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = b0 + j < n_part ? part[(size_t)(b0 + j) * nacc + k] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[j];
    }
    S.acc[k] = a;
}

struct UpdateArgs {
    int n_part;        // workgroups of the residual launch whose partial sums are added (in order) first
    int do_backward;   // 0: only build the camera table (first call)
    int stage;         // trainable set: 1 = quats, trans, log_sizes ; 2 = + pps, log_focals
    float lr;          // cosine-scheduled learning rate of this step
    int step;          // 1-based Adam step within the stage
    int loss_index;    // where to store the loss of the step just evaluated
    int reset_moments; // first step of a stage: fresh optimiser (reconstruct.py:374)
    int opt_pp;        // stage 2 also moves the principal points (reconstruct.py:436)
    float step_size;   // lr / (1 - 0.9^step) and sqrt(1 - 0.9^step): Adam's bias corrections, evaluated in double on the host
    float bc2_sqrt;    // (a double-precision pow per thread used to sit on the single workgroup's critical path)
};

// The chain walks are sequential over the MST edges, but the 12 (forward) / 24 (reverse) numbers of one edge are
// independent: the lanes of wave 0 compute one each, so an edge costs one LDS round trip instead of ~150 dependent
// instructions of a single thread (this kernel is one workgroup: its time is its longest dependent chain).
// LDS operations of one wave complete in order; the wave barrier only stops the compiler from reordering them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// kinematic chain over the MST (reconstruct.py:233-238): Rt_b = Rt_a Rr_b, tt_b = Rt_a tr_b + tt_a.
// Called by the 64 lanes of wave 0.
__device__ __forceinline__ void chain_forward_wave(int lane, int root, int n_edges, const int* sedge, const float* sRr,
                                                   const float* strans, float* sRt, float* stt) {
    if (lane < 9) sRt[9 * root + lane] = sRr[9 * root + lane];
    else if (lane < 12) stt[3 * root + lane - 9] = strans[3 * root + lane - 9];
    wave_lds_sync();
    for (int e = 0; e < n_edges; ++e) {
        const int a = sedge[2 * e], b = sedge[2 * e + 1];
        float val = 0.f;
        if (lane < 9) {
            const int r = lane / 3, c = lane - 3 * r;
            val = sRt[9 * a + 3 * r] * sRr[9 * b + c] + sRt[9 * a + 3 * r + 1] * sRr[9 * b + 3 + c] +
                  sRt[9 * a + 3 * r + 2] * sRr[9 * b + 6 + c];
        } else if (lane < 12) {
            const int r = lane - 9;
            val = sRt[9 * a + 3 * r] * strans[3 * b] + sRt[9 * a + 3 * r + 1] * strans[3 * b + 1] +
                  sRt[9 * a + 3 * r + 2] * strans[3 * b + 2] + stt[3 * a + r];
        }
        wave_lds_sync();
        if (lane < 9) sRt[9 * b + lane] = val;
        else if (lane < 12) stt[3 * b + lane - 9] = val;
        wave_lds_sync();
    }
}

// reverse walk: gradients of the chained poses -> gradients of the relative poses (in place in svRt / svtt)
__device__ __forceinline__ void chain_reverse_wave(int lane, int n_edges, const int* sedge, const float* sRr,
                                                   const float* strans, const float* sRt, float* svRt, float* svtt) {
    for (int e = n_edges - 1; e >= 0; --e) {
        const int a = sedge[2 * e], b = sedge[2 * e + 1];
        float val = 0.f;
        if (lane < 9) {            // Rt_b = Rt_a Rr_b : vRt_a += vRt_b Rr_b^T ; tt_b = Rt_a tr_b + tt_a : vRt_a += vtt_b (x) tr_b
            const int r = lane / 3, c = lane - 3 * r;
            val = svRt[9 * a + lane] + svRt[9 * b + 3 * r] * sRr[9 * b + 3 * c] + svRt[9 * b + 3 * r + 1] * sRr[9 * b + 3 * c + 1] +
                  svRt[9 * b + 3 * r + 2] * sRr[9 * b + 3 * c + 2];
            val += svtt[3 * b + r] * strans[3 * b + c];
        } else if (lane < 18) {    // vRr_b = Rt_a^T vRt_b
            const int k = lane - 9, r = k / 3, c = k - 3 * r;
            val = sRt[9 * a + r] * svRt[9 * b + c] + sRt[9 * a + 3 + r] * svRt[9 * b + 3 + c] + sRt[9 * a + 6 + r] * svRt[9 * b + 6 + c];
        } else if (lane < 21) {    // vtr_b = Rt_a^T vtt_b
            const int r = lane - 18;
            val = sRt[9 * a + r] * svtt[3 * b] + sRt[9 * a + 3 + r] * svtt[3 * b + 1] + sRt[9 * a + 6 + r] * svtt[3 * b + 2];
        } else if (lane < 24) {    // vtt_a += vtt_b
            const int r = lane - 21;
            val = svtt[3 * a + r] + svtt[3 * b + r];
        }
        wave_lds_sync();           // every lane has read its inputs before anything is overwritten
        if (lane < 9) svRt[9 * a + lane] = val;
        else if (lane < 18) svRt[9 * b + lane - 9] = val;   // from here on svRt[b] / svtt[b] belong to the RELATIVE pose of b
        else if (lane < 21) svtt[3 * b + lane - 18] = val;
        else if (lane < 24) svtt[3 * a + lane - 21] = val;
        wave_lds_sync();
    }
}

// single workgroup; thread i < C owns view i for the element-wise parts, thread 0 walks the chain
#ifdef ALIGN_PROFILE   // checkpoints of the update phase (thread 0, shader clock): sums over all calls -> g_upd_prof
__device__ unsigned long long g_upd_prof[8];
#define UPD_MARK(k) do { if (threadIdx.x == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); \
                         atomicAdd(&g_upd_prof[k], (unsigned long long)(t_ - upd_t)); upd_t = t_; } } while (0)
#else
#define UPD_MARK(k) do { } while (0)
#endif

__device__ __forceinline__ void align_update_body(const AlignProblem& P, const AlignState& S, const UpdateArgs& U) {
    __shared__ float sRr[MAXC * 9], sRt[MAXC * 9], stt[MAXC * 3];        // relative / chained rotations, chained translation
    __shared__ float svRt[MAXC * 9], svtt[MAXC * 3];                      // their gradients
    __shared__ float ssize[MAXC], svgs[MAXC];
    __shared__ float strans[MAXC * 3];   // relative translations and the MST edges, staged once: the three serial chain
    __shared__ int sedge[MAXC * 2];      // walks below are done by one thread and must not wait on global memory
    __shared__ float s_gs, s_vgs, s_min; __shared__ int s_argmin;
    const int i = threadIdx.x;
    const int C = P.C;
#ifdef ALIGN_PROFILE
    long long upd_t = (long long)__builtin_readcyclecounter();
#endif
    for (int k = i; k < 3 * C; k += blockDim.x) strans[k] = S.trans[k];
    for (int k = i; k < 2 * P.n_edges; k += blockDim.x) sedge[k] = P.edges[k];
    float* flags = S.acc + C * ACC_STRIDE;
    if (U.do_backward && *(flags + 1) != 0.f) return;  // stopped earlier by a NaN loss
    if (U.do_backward && U.n_part >= 0) {   // gradient sums and loss = the residual workgroups' partials, added in
                                            // workgroup order (n_part < 0: k_align_reduce has done it)
        const int nacc = C * ACC_STRIDE + 1;
        const float* __restrict__ part = S.part;
        // 32 loads in flight per thread (one round trip for up to 32 workgroups), added strictly in workgroup order
        constexpr int INFL = 32;
        for (int k = i; k < nacc; k += blockDim.x) {
            float a = 0.f;
            for (int b0 = 0; b0 < U.n_part; b0 += INFL) {
                float v[INFL];
#pragma unroll
                for (int j = 0; j < INFL; ++j) v[j] = b0 + j < U.n_part ? *(part + (size_t)(b0 + j) * nacc + k) : 0.f;
#pragma unroll
                for (int j = 0; j < INFL; ++j) a += v[j];
            }
            S.acc[k] = a;
        }
        __syncthreads();
        UPD_MARK(0);
    }
    if (U.do_backward && i == 0) {
        const float loss = flags[0];
        S.losses[U.loss_index] = loss;
        if (loss != loss) flags[1] = 1.f;          // reference: `if loss != loss: break` (:397-399), before... after the step
    }
    if (U.reset_moments)
        for (int k = i; k < 11 * C; k += blockDim.x) { S.m[k] = 0.f; S.v[k] = 0.f; }
    __syncthreads();
    UPD_MARK(1);

    float pv[11];          // this view's parameters (pps 2, log_focal, quat 4, trans 3, log_size) once they are in registers
    bool have_pv = false;
    // ---------------- backward of make_K_cam_depth + Adam (uses the forward state of the step just evaluated) ----------------
    if (U.do_backward) {
        // recompute the forward pieces this thread needs
        float f = 0, s = 0, zc = 0, W = 0, H = 0, ppx = 0, ppy = 0, med = 0, bf = 0, to[3] = {0, 0, 0};
        bool fclip = false;
        if (i < C) {
            W = P.imsizes[2 * i]; H = P.imsizes[2 * i + 1];
            const float fe = __expf(S.log_focals[i]);
            f = fminf(fmaxf(fe, P.min_focals[i]), P.max_focals[i]);
            fclip = (fe < P.min_focals[i]) || (fe > P.max_focals[i]);
            s = __expf(S.log_sizes[i]); ssize[i] = s;
            med = P.median[i]; bf = P.base_focals[i];
            zc = s * med * f / bf;
            ppx = S.pps[2 * i]; ppy = S.pps[2 * i + 1];
            to[0] = zc * (W / f) * (0.5f - ppx); to[1] = zc * (H / f) * (0.5f - ppy); to[2] = zc;
            float qn[4], inv;
            quat_to_rot(S.quats + 4 * i, sRr + 9 * i, qn, &inv);
        }
        __syncthreads();
        if (i == 0) {
            // torch's full-reduction min() splits its gradient evenly between tied minima (all sizes are
            // exactly 1 on the very first step), so count the ties
            float mn = ssize[0];
            for (int k = 1; k < C; ++k) mn = fminf(mn, ssize[k]);
            int ties = 0;
            for (int k = 0; k < C; ++k) ties += (ssize[k] == mn) ? 1 : 0;
            s_gs = 1.0f / mn; s_argmin = ties; s_vgs = 0.f; s_min = mn;
        }
        // chained rotations + translations of the current parameters: the forward half of the previous call left them
        // (one parallel load instead of a sequential walk over the MST: ~0.45 us per edge on this one-workgroup kernel)
        if (S.chain) {
            for (int k = i; k < 9 * C; k += blockDim.x) sRt[k] = S.chain[k];
            for (int k = i; k < 3 * C; k += blockDim.x) stt[k] = S.chain[9 * C + k];
        } else if (i < 64) chain_forward_wave(i, P.root, P.n_edges, sedge, sRr, strans, sRt, stt);
        __syncthreads();
        UPD_MARK(2);
        const float gs = s_gs;
        float v_f = 0, v_ppx = 0, v_ppy = 0, v_s = 0, vgs_part = 0;
        if (i < C) {
            float g[17];
            for (int k = 0; k < 17; ++k) g[k] = S.acc[i * ACC_STRIDE + k];
            const float* Rt = sRt + 9 * i;
            const float vT[3] = {g[9], g[10], g[11]};
            const float vA = g[15], vB = g[16];
            // T = gs (tt - Rt to) ; A = gs (zc - med s) ; B = gs med s
            float Rto[3];
            for (int r = 0; r < 3; ++r) Rto[r] = Rt[3 * r] * to[0] + Rt[3 * r + 1] * to[1] + Rt[3 * r + 2] * to[2];
            for (int r = 0; r < 3; ++r) vgs_part += vT[r] * (stt[3 * i + r] - Rto[r]);
            vgs_part += vA * (zc - med * s) + vB * med * s;
            float vto[3];
            for (int c = 0; c < 3; ++c) vto[c] = -gs * (Rt[c] * vT[0] + Rt[3 + c] * vT[1] + Rt[6 + c] * vT[2]);
            for (int r = 0; r < 3; ++r) {
                svtt[3 * i + r] = gs * vT[r];
                for (int c = 0; c < 3; ++c) svRt[9 * i + 3 * r + c] = g[3 * r + c] - gs * vT[r] * to[c];
            }
            const float ax = (W / f) * (0.5f - ppx), ay = (H / f) * (0.5f - ppy);
            const float v_zc = vA * gs + vto[0] * ax + vto[1] * ay + vto[2];
            v_f = g[12] + vto[0] * zc * (-(W / (f * f)) * (0.5f - ppx)) + vto[1] * zc * (-(H / (f * f)) * (0.5f - ppy)) +
                  v_zc * s * med / bf;
            v_ppx = g[13] * W - vto[0] * zc * (W / f);
            v_ppy = g[14] * H - vto[1] * zc * (H / f);
            v_s = v_zc * med * f / bf - vA * gs * med + vB * gs * med;
            svgs[i] = vgs_part;
        }
        __syncthreads();
        if (i == 0) {   // fixed order (an LDS float atomic per camera left the order to the hardware)
            float a = 0.f;
            for (int k = 0; k < C; ++k) a += svgs[k];
            s_vgs = a;
        }
        __syncthreads();
        if (i < 64) chain_reverse_wave(i, P.n_edges, sedge, sRr, strans, sRt, svRt, svtt);
        __syncthreads();
        UPD_MARK(3);
        if (i < C) {
            if (s == s_min) v_s += s_vgs * (-gs * gs) / (float)s_argmin;  // gs = 1/min(s); s_argmin = tie count
            // quaternion (x,y,z,w): rotmat VJP, then the normalisation VJP
            float R[9], qn[4], inv;
            quat_to_rot(S.quats + 4 * i, R, qn, &inv);
            const float* vR = svRt + 9 * i;
            const float x = qn[0], y = qn[1], z = qn[2], w = qn[3];
            float vq[4];
            vq[3] = 2.0f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
            vq[0] = 2.0f * (-2.0f * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
            vq[1] = 2.0f * (x * (vR[3] + vR[1]) - 2.0f * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
            vq[2] = 2.0f * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - 2.0f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
            const float dq = vq[0] * x + vq[1] * y + vq[2] * z + vq[3] * w;
            for (int k = 0; k < 4; ++k) vq[k] = (vq[k] - dq * qn[k]) * inv;
            // gradients in the optimiser's parameter order
            float grad[11];
            grad[0] = v_ppx; grad[1] = v_ppy; grad[2] = fclip ? 0.f : v_f * f;
            for (int k = 0; k < 4; ++k) grad[3 + k] = vq[k];
            for (int k = 0; k < 3; ++k) grad[7 + k] = svtt[3 * i + k];
            grad[10] = v_s * s;
            float* pptr[11] = {S.pps + 2 * i, S.pps + 2 * i + 1, S.log_focals + i, S.quats + 4 * i, S.quats + 4 * i + 1,
                               S.quats + 4 * i + 2, S.quats + 4 * i + 3, S.trans + 3 * i, S.trans + 3 * i + 1,
                               S.trans + 3 * i + 2, S.log_sizes + i};
            const int moff[11] = {2 * i, 2 * i + 1, 2 * C + i, 3 * C + 4 * i, 3 * C + 4 * i + 1, 3 * C + 4 * i + 2,
                                  3 * C + 4 * i + 3, 7 * C + 3 * i, 7 * C + 3 * i + 1, 7 * C + 3 * i + 2, 10 * C + i};
            // Adam(lr, betas=(0.9, 0.9), eps=1e-8), torch single-tensor semantics
            const float step_size = U.step_size, bc2_sqrt = U.bc2_sqrt;
            const float w1 = (float)(1.0 - 0.9), b2 = 0.9f, eps = 1e-8f;
            // the eleven parameters of the view stay in registers from here to the end of the call (pv): no store ->
            // load round trips on this one-workgroup kernel's critical path
#pragma unroll
            for (int k = 0; k < 11; ++k) pv[k] = *pptr[k];
            have_pv = true;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const bool trainable = (k >= 3) || (U.stage == 2 && (k == 2 || U.opt_pp));
                if (!trainable) continue;
                const float gk = grad[k];
                float mk = S.m[moff[k]], vk = S.v[moff[k]];
                mk = fmaf(w1, gk - mk, mk);
                vk = vk * b2 + (w1 * gk) * gk;
                S.m[moff[k]] = mk; S.v[moff[k]] = vk;
                pv[k] = pv[k] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
            }
            // make sure the pose remains well optimizable (reconstruct.py:394-395)
            const float n = sqrtf(pv[3] * pv[3] + pv[4] * pv[4] + pv[5] * pv[5] + pv[6] * pv[6]);
            pv[3] /= n; pv[4] /= n; pv[5] /= n; pv[6] /= n;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const bool trainable = (k >= 3) || (U.stage == 2 && (k == 2 || U.opt_pp));
                if (trainable) *pptr[k] = pv[k];
            }
            for (int k = 0; k < 3; ++k) strans[3 * i + k] = pv[7 + k];
        }
        __syncthreads();
        UPD_MARK(4);
    }

    // ---------------- forward: camera table for the next residual launch ----------------
    if (i < C) {
        if (!have_pv) {
            pv[0] = S.pps[2 * i]; pv[1] = S.pps[2 * i + 1]; pv[2] = S.log_focals[i];
            for (int k = 0; k < 4; ++k) pv[3 + k] = S.quats[4 * i + k];
            for (int k = 0; k < 3; ++k) pv[7 + k] = S.trans[3 * i + k];
            pv[10] = S.log_sizes[i];
        }
        const float s = __expf(pv[10]);
        ssize[i] = s;
        float qn[4], inv;
        quat_to_rot(pv + 3, sRr + 9 * i, qn, &inv);
    }
    __syncthreads();
    if (i == 0) {
        float mn = ssize[0];
        for (int k = 1; k < C; ++k) mn = fminf(mn, ssize[k]);
        s_gs = 1.0f / mn;
    }
    if (i < 64) chain_forward_wave(i, P.root, P.n_edges, sedge, sRr, strans, sRt, stt);
    __syncthreads();
    UPD_MARK(5);
    if (S.chain) {
        for (int k = i; k < 9 * C; k += blockDim.x) S.chain[k] = sRt[k];
        for (int k = i; k < 3 * C; k += blockDim.x) S.chain[9 * C + k] = stt[k];
    }
    if (i < C) {
        const float gs = s_gs;
        const float W = P.imsizes[2 * i], H = P.imsizes[2 * i + 1];
        const float f = fminf(fmaxf(__expf(pv[2]), P.min_focals[i]), P.max_focals[i]);
        const float s = ssize[i], med = P.median[i], bf = P.base_focals[i];
        const float zc = s * med * f / bf;
        const float ppx = pv[0], ppy = pv[1];
        const float to[3] = {zc * (W / f) * (0.5f - ppx), zc * (H / f) * (0.5f - ppy), zc};
        float* c = S.cam + i * CAM_STRIDE;
        const float* Rt = sRt + 9 * i;
        for (int k = 0; k < 9; ++k) c[k] = Rt[k];
        for (int r = 0; r < 3; ++r)
            c[9 + r] = gs * (stt[3 * i + r] - (Rt[3 * r] * to[0] + Rt[3 * r + 1] * to[1] + Rt[3 * r + 2] * to[2]));
        c[12] = f; c[13] = ppx * W; c[14] = ppy * H;
        c[15] = gs * (zc - med * s); c[16] = gs * med * s; c[17] = bf;
    }
    // clear the accumulators for the next residual launch (keep the NaN flag)
    for (int k = i; k < C * ACC_STRIDE + 1; k += blockDim.x) S.acc[k] = 0.f;
}

__global__ __launch_bounds__(256) void k_align_update(AlignProblem P, AlignState S, UpdateArgs U) {
    align_update_body(P, S, U);
}

// opt_depth (reconstruct.py:437): the core depths are parameters of the second stage as well.  One thread per core
// depth adds the numbers of its rows in the fixed order of the caller's grouping (csr_off / csr_rows: rows of stage 2
// -- loss_2d rows, then regression rows -- grouped by the core depth their anchor reads), then Adam(0.9, 0.9) like
// k_align_update.  No float atomics: bit-reproducible.
__global__ __launch_bounds__(256) void k_align_depth_update(int n_elem, float* __restrict__ core, float* __restrict__ m,
                                                            float* __restrict__ v, const int32_t* __restrict__ csr_off,
                                                            const int32_t* __restrict__ csr_rows,
                                                            const float* __restrict__ rowgrad,
                                                            const float* __restrict__ flags, float lr, int step,
                                                            int reset_moments) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem || flags[1] != 0.f) return;   // (stopped earlier by a NaN loss)
    float g = 0.f;
    for (int j = csr_off[e]; j < csr_off[e + 1]; ++j) g += rowgrad[csr_rows[j]];
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc1);
    const float w1 = (float)(1.0 - 0.9), b2 = 0.9f, eps = 1e-8f;
    float mk = reset_moments ? 0.f : m[e], vk = reset_moments ? 0.f : v[e];
    mk = fmaf(w1, g - mk, mk);
    vk = vk * b2 + (w1 * g) * g;
    m[e] = mk; v[e] = vk;
    core[e] = core[e] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
}

// world points of every anchor from the current camera table (the reference's `pts3d` result, :405-406)
__global__ void k_align_points(AlignProblem P, AlignState S, int n_anchors, float* __restrict__ pts) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_anchors) return;
    const Pt r = anchor_point(P, S.cam, a);
    pts[3 * a] = r.pw[0]; pts[3 * a + 1] = r.pw[1]; pts[3 * a + 2] = r.pw[2];
}

#ifdef ALIGN_PROFILE
ST3R_EXPORT int st3r_debug_update_profile(unsigned long long* out8_host, int reset) {
    if (out8_host) (void)hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_upd_prof), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_upd_prof), z, sizeof(z)); }
    return 0;
}
#endif

static float gamma_offset(float gamma) {
    return gamma == 1.0f ? 0.0f : (float)pow(1.0 / (double)gamma, 1.0 / ((double)gamma - 1.0));
}

ST3R_EXPORT int st3r_align_run_opts(st3r_ctx* ctx, void* stream, int C, int G, int n_anchors, const float* imsizes,
                               const float* base_focals, const float* median, float* core,
                               const float* min_focals, const float* max_focals, const float* anchor_pix,
                               const int32_t* anchor_idx, const float* anchor_off, const int32_t* anchor_img,
                               int n_corr, const int32_t* corr_a1, const int32_t* corr_a2, const float* corr_w,
                               int n_c2d, const float* c2d_pix, const int32_t* c2d_a2, const int32_t* c2d_img1,
                               const float* c2d_w, int n_dust, const int32_t* dust_a1, const float* dust_tgt,
                               const int32_t* dust_img2, const float* dust_w, int root, int n_edges,
                               const int32_t* edges, float lr1, int niter1, float lr2, int niter2, float dust_weight,
                               float* pps, float* log_focals, float* quats, float* trans, float* log_sizes,
                               float* work, int64_t work_floats, float* cam_out, float* pts_out, float* losses_out,
                               const float* lr_host, float gamma1, float gamma2, float gammad, int opt_pp,
                               const int32_t* depth_csr_off, const int32_t* depth_csr_rows, float* depth_work,
                               int64_t depth_work_floats) {
    ARG_CHECK(ctx && C > 0 && C <= MAXC && G > 0 && n_anchors >= 0 && niter1 >= 0 && niter2 >= 0);
    ARG_CHECK(gamma1 > 0.f && gamma2 > 0.f && gammad > 0.f);
    const bool opt_depth = depth_csr_off != nullptr;
    const int64_t n_core = (int64_t)C * G, n_rows2 = (int64_t)n_c2d + n_dust;
    ARG_CHECK(!opt_depth || (depth_work && depth_work_floats >= n_rows2 + 3 * n_core && (n_rows2 == 0 || depth_csr_rows)));
    ARG_CHECK(imsizes && base_focals && median && core && min_focals && max_focals && pps && log_focals && quats &&
              trans && log_sizes && work && cam_out && edges);
    ARG_CHECK(n_edges == C - 1 && root >= 0 && root < C);
    const int64_t need = 22 * (int64_t)C + (int64_t)C * CAM_STRIDE + (int64_t)C * ACC_STRIDE + 8;
    ARG_CHECK(work_floats >= need);
    hipStream_t s = (hipStream_t)stream;
    AlignProblem P;
    P.C = C; P.imsizes = imsizes; P.base_focals = base_focals; P.median = median; P.core = core; P.G = G;
    P.anchor_pix = anchor_pix; P.anchor_idx = anchor_idx; P.anchor_off = anchor_off; P.anchor_img = anchor_img;
    P.n_corr = n_corr; P.corr_a1 = corr_a1; P.corr_a2 = corr_a2; P.corr_w = corr_w;
    P.n_c2d = n_c2d; P.c2d_pix = c2d_pix; P.c2d_a2 = c2d_a2; P.c2d_img1 = c2d_img1; P.c2d_w = c2d_w;
    P.n_dust = n_dust; P.dust_a1 = dust_a1; P.dust_tgt = dust_tgt; P.dust_img2 = dust_img2; P.dust_w = dust_w;
    P.root = root; P.n_edges = n_edges; P.edges = edges; P.min_focals = min_focals; P.max_focals = max_focals;
    P.gamma1 = gamma1; P.off1 = gamma_offset(gamma1); P.gamma2 = gamma2; P.off2 = gamma_offset(gamma2);
    P.gammad = gammad; P.offd = gamma_offset(gammad);
    // opt_depth scratch: per-row numbers | Adam moments of the core depths | the core depths as they were when the
    // results were exported (the reference's depthmaps are one step behind its parameters, see export_results)
    P.rowgrad = opt_depth ? depth_work : nullptr;
    float* core_m = opt_depth ? depth_work + n_rows2 : nullptr;
    float* core_v = opt_depth ? core_m + n_core : nullptr;
    float* core_snapshot = opt_depth ? core_v + n_core : nullptr;
    AlignState S;
    S.pps = pps; S.log_focals = log_focals; S.quats = quats; S.trans = trans; S.log_sizes = log_sizes;
    S.m = work; S.v = work + 11 * C; S.cam = work + 22 * C; S.acc = S.cam + (int64_t)C * CAM_STRIDE;
    S.losses = losses_out;
    {   // the chain cache of k_align_update: 12 floats per view
        void* pc;
        int rc = st3r_arena_get(ctx, SLOT_ALIGN_CTL, sizeof(float) * (size_t)(12 * C), &pc);
        if (rc) return rc;
        S.chain = (float*)pc;
    }
    auto lr_of = [&](int stage, int it, int li) -> float {
        const int niter = stage == 1 ? niter1 : niter2;
        const float lr_base = stage == 1 ? lr1 : lr2;
        return lr_host ? lr_host[li]   // the caller's schedule(alpha, lr_base, lr_end), evaluated per iteration
                       : (float)(0.0 + ((double)lr_base - 0.0) * (1.0 + cos(((double)it / niter) * M_PI)) / 2.0);  // cosine_schedule
    };
    auto adam_factors = [](float lr, int step, float* step_size, float* bc2_sqrt) {   // Adam(betas = (0.9, 0.9)), torch
        const double bc1 = 1.0 - pow(0.9, (double)step);
        *step_size = (float)((double)lr / bc1); *bc2_sqrt = (float)sqrt(bc1);
    };
    HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * (size_t)need, s));
    const size_t sh = sizeof(float) * (4 * ((size_t)C * ACC_STRIDE + 1) + (size_t)C * CAM_STRIDE);
    if (sh > 64 * 1024)   // more than ~190 views: the four accumulator copies pass the default dynamic-LDS limit
        HIP_TRY(hipFuncSetAttribute((const void*)k_align_resid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    // one residual workgroup per 256 rows up to 1024 workgroups (then `rpt` groups of 256 rows each); up to 32 partials
    // are added by k_align_update itself, more by k_align_reduce (one thread per accumulator word)
    const int max_rows = (n_corr > n_c2d ? n_corr : n_c2d) + n_dust;
    const int rpt = max_rows > 0 ? ceil_div(ceil_div(max_rows, 256), 1024) : 1;
    {
        void* pp;
        int rc = st3r_arena_get(ctx, SLOT_SCAN_TMP, sizeof(float) * 1024 * ((size_t)C * ACC_STRIDE + 1), &pp);
        if (rc) return rc;
        S.part = (float*)pp;
    }
    {   // constants of every anchor, packed once so that the residual kernel has a single dependent load level
        void* pk;
        int rc = st3r_arena_get(ctx, SLOT_NN_PART, sizeof(float4) * 2 * (size_t)(n_anchors > 0 ? n_anchors : 1), &pk);
        if (rc) return rc;
        P.anchor_pack = (const float4*)pk;
        if (n_anchors > 0)
            hipLaunchKernelGGL(k_align_pack_anchors, dim3(ceil_div(n_anchors, 256)), dim3(256), 0, s, n_anchors, anchor_pix,
                               anchor_idx, anchor_off, anchor_img, core, G, (float4*)pk);
    }
    UpdateArgs U0 = {0, 0, 1, 0.f, 0, 0, 0, 1, 0.f, 1.f};
    hipLaunchKernelGGL(k_align_update, dim3(1), dim3(256), 0, s, P, S, U0);
    // The reference returns K / cam2w / depthmaps / pts3d as computed at the START of the last iteration,
    // i.e. one optimiser step behind the returned parameters (optimize_loop builds them before
    // loss.backward()/step and never refreshes them, reconstruct.py:379-380,405-406).  Reproduced here:
    // the camera table is exported right before the final update launch.
    auto export_results = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(cam_out, S.cam, sizeof(float) * (size_t)C * CAM_STRIDE, hipMemcpyDeviceToDevice, s));
        if (opt_depth)
            HIP_TRY(hipMemcpyAsync(core_snapshot, core, sizeof(float) * (size_t)n_core, hipMemcpyDeviceToDevice, s));
        if (pts_out && n_anchors > 0)
            hipLaunchKernelGGL(k_align_points, dim3(ceil_div(n_anchors, 256)), dim3(256), 0, s, P, S, n_anchors, pts_out);
        return ST3R_OK;
    };
    const int last_stage = niter2 > 0 ? 2 : 1;
    if (niter1 == 0 && niter2 == 0) { int rc = export_results(); if (rc) return rc; }
    int li = 0;
    for (int stage = 1; stage <= 2; ++stage) {
        const int niter = stage == 1 ? niter1 : niter2;
        const int rows = (stage == 1 ? n_corr : n_c2d) + n_dust;
        for (int it = 0; it < niter; ++it) {
            if (stage == last_stage && it == niter - 1) { int rc = export_results(); if (rc) return rc; }
            const int n_part = rows > 0 ? ceil_div(rows, 256 * rpt) : 0;
            if (rows > 0) hipLaunchKernelGGL(k_align_resid, dim3(n_part), dim3(256), sh, s, P, S, stage, dust_weight, rpt);
            const int nacc = C * ACC_STRIDE + 1;
            if (n_part > 32) hipLaunchKernelGGL(k_align_reduce, dim3(ceil_div(nacc, 256)), dim3(256), 0, s, S, nacc, n_part);
            UpdateArgs U;
            U.n_part = n_part > 32 ? -1 : n_part; U.do_backward = 1; U.stage = stage;
            U.lr = lr_of(stage, it, li);
            adam_factors(U.lr, it + 1, &U.step_size, &U.bc2_sqrt);
            U.step = it + 1; U.loss_index = li++; U.reset_moments = (it == 0); U.opt_pp = opt_pp;
            const bool depth_step = opt_depth && stage == 2 && rows > 0;
            if (depth_step)   // before k_align_update: both then see the NaN flag of the EARLIER iterations
                hipLaunchKernelGGL(k_align_depth_update, dim3(ceil_div((int)n_core, 256)), dim3(256), 0, s, (int)n_core, core,
                                   core_m, core_v, depth_csr_off, depth_csr_rows, P.rowgrad, S.acc + C * ACC_STRIDE, U.lr,
                                   U.step, U.reset_moments);
            hipLaunchKernelGGL(k_align_update, dim3(1), dim3(256), 0, s, P, S, U);
            if (depth_step && n_anchors > 0)   // the anchors' packed core depths follow the parameters
                hipLaunchKernelGGL(k_align_pack_anchors, dim3(ceil_div(n_anchors, 256)), dim3(256), 0, s, n_anchors,
                                   anchor_pix, anchor_idx, anchor_off, anchor_img, core, G, (float4*)P.anchor_pack);
        }
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}

// The reference's own configuration (reconstruct.py:118-122 defaults): gamma losses 1.1 / 0.4 / 1.1, cosine schedule, opt_pp.
ST3R_EXPORT int st3r_align_run(st3r_ctx* ctx, void* stream, int C, int G, int n_anchors, const float* imsizes,
                               const float* base_focals, const float* median, const float* core,
                               const float* min_focals, const float* max_focals, const float* anchor_pix,
                               const int32_t* anchor_idx, const float* anchor_off, const int32_t* anchor_img,
                               int n_corr, const int32_t* corr_a1, const int32_t* corr_a2, const float* corr_w,
                               int n_c2d, const float* c2d_pix, const int32_t* c2d_a2, const int32_t* c2d_img1,
                               const float* c2d_w, int n_dust, const int32_t* dust_a1, const float* dust_tgt,
                               const int32_t* dust_img2, const float* dust_w, int root, int n_edges,
                               const int32_t* edges, float lr1, int niter1, float lr2, int niter2, float dust_weight,
                               float* pps, float* log_focals, float* quats, float* trans, float* log_sizes,
                               float* work, int64_t work_floats, float* cam_out, float* pts_out, float* losses_out) {
    return st3r_align_run_opts(ctx, stream, C, G, n_anchors, imsizes, base_focals, median, (float*)core, min_focals, max_focals,
                               anchor_pix, anchor_idx, anchor_off, anchor_img, n_corr, corr_a1, corr_a2, corr_w, n_c2d,
                               c2d_pix, c2d_a2, c2d_img1, c2d_w, n_dust, dust_a1, dust_tgt, dust_img2, dust_w, root,
                               n_edges, edges, lr1, niter1, lr2, niter2, dust_weight, pps, log_focals, quats, trans,
                               log_sizes, work, work_floats, cam_out, pts_out, losses_out, nullptr, 1.1f, 0.4f, 1.1f, 1,
                               nullptr, nullptr, nullptr, 0);
}
