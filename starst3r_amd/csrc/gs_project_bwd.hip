// K1-bwd: backward of projection + SH colour, one thread per Gaussian looping over the
// cameras.  The per-camera gradients of the world mean and world covariance are summed in
// registers, the quaternion/scale VJP runs once per Gaussian, and the 23 gradients are
// written exactly once -- no atomics, no read-modify-write of the gradient buffer.
//
// Replaces gsplat fully_fused_projection_packed_bwd + spherical_harmonics bwd (autograd
// through starster/gs.py:76-87 from loss.backward(), starster/gs.py:153) and folds in the
// gradients of the two regularisers of starster/gs.py:132-134.
//
// Round 5, GATHER = true (the fused training calls): the kernel also does what k_gather_vtile (gs_blend.hip) did in a launch
// of its own -- summing a (camera, Gaussian) pair's stamped (record, tile) slots in slot order.  A workgroup's 256 Gaussians
// are, for one camera, 256 consecutive pairs -- a wave's 64 one contiguous slot range: the same cooperative pattern (lane = slot
// through LDS, then lane = pair adds its own rows, the next slots in flight), per wave and camera, and the nine sums go straight
// into the chain rule instead of through 48 bytes per pair of HBM each way (0.58 GB per step at SYNTH-1M).  Same sums in the
// same order: bit-identical gradients.
#include "common.h"
#include "blend_common.h"

#define CAM_STRIDE 32
#ifndef GATHER_WIDE_ABOVE
#define GATHER_WIDE_ABOVE 6   // slots per pair (wave average) above which the slot gather deals its additions by item
#endif
#define SH_C0 0.2820947917738781f
#define SH_C1 0.48860251190292f

template <bool GATHER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_project_sh_bwd(
    int N, int C, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ sh,
    int sh_stride, const float* __restrict__ viewmats, const float* __restrict__ Ks,
    const float* __restrict__ campos, int W, int H, float eps2d, const float4* __restrict__ splats,
    const float4* __restrict__ v_splats, float reg_o_k, float reg_s_k, float* __restrict__ grads, int accumulate,
    int g_begin, int g_end, int range_major, const int32_t* __restrict__ cum, const float* __restrict__ vtile, int stamp,
    unsigned vt_cap, const uint32_t* __restrict__ touch) {
    extern __shared__ float cam[];
    constexpr int ROW = ACC_VALS;   // odd stride: rows of neighbouring slots fall into different banks
    __shared__ float sVal[GATHER ? 256 * ROW : 1];    // per wave: the current chunk of 64 slots, nine values each
    __shared__ float sAcc[GATHER ? 256 * ROW : 1];    // per wave: the running sums of its 64 pairs
    __shared__ int sStart[GATHER ? 4 * 65 : 1];       // per wave: first slot of each of its pairs (+ the end of the last)
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
    const int gbase = g_begin + blockIdx.x * blockDim.x;   // one launch per Gaussian range (see comm.hip)
    const bool valid = gbase + (int)threadIdx.x < g_end;
    if (!GATHER && !valid) return;
    const int g = valid ? gbase + (int)threadIdx.x : g_end - 1;   // (GATHER: every lane of a wave takes part in its gather)

    const float mx = means[3 * g], my = means[3 * g + 1], mz = means[3 * g + 2];
    const float opac = opacities[g];
    float qw = quats[4 * g], qx = quats[4 * g + 1], qy = quats[4 * g + 2], qz = quats[4 * g + 3];
    const float s0 = scales[3 * g], s1 = scales[3 * g + 1], s2 = scales[3 * g + 2];
    float k[12];
    const float* kp = sh + (int64_t)g * sh_stride;
#pragma unroll
    for (int i = 0; i < 12; ++i) k[i] = kp[i];
    const float inv_norm = 1.0f / sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    qw *= inv_norm; qx *= inv_norm; qy *= inv_norm; qz *= inv_norm;
    // rotation and M = Rq diag(s): needed for the covariance here and for the quaternion / scale chain rule after the camera
    // loop -- computed twice rather than kept in 18 registers across the loop (the fused kernel is short of them)
    const float sc[3] = {s0, s1, s2};
    auto rot_and_m = [&](float (&Rq)[9], float (&M)[9]) {
        float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
        float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        Rq[0] = 1.0f - 2.0f * (y2 + z2); Rq[1] = 2.0f * (xy - wz); Rq[2] = 2.0f * (xz + wy);
        Rq[3] = 2.0f * (xy + wz); Rq[4] = 1.0f - 2.0f * (x2 + z2); Rq[5] = 2.0f * (yz - wx);
        Rq[6] = 2.0f * (xz - wy); Rq[7] = 2.0f * (yz + wx); Rq[8] = 1.0f - 2.0f * (x2 + y2);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * sc[j];
    };
    float cov[6];
    {
        float Rq[9], M[9];
        rot_and_m(Rq, M);
        cov[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
        cov[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
        cov[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
        cov[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
        cov[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
        cov[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
        asm volatile("" : "+v"(cov[0]), "+v"(cov[1]), "+v"(cov[2]), "+v"(cov[3]), "+v"(cov[4]), "+v"(cov[5]));
    }

    float v_mean[3] = {0, 0, 0};
    float vSw[6] = {0, 0, 0, 0, 0, 0};  // symmetric world-covariance gradient: 00 01 02 11 12 22
    float v_k[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v_k[i] = 0.f;
    float v_opac = 0.f;

    for (int c = 0; c < C; ++c) {
        const int64_t pid = (int64_t)c * N + g;
        float4 g0, g1, g2;
        if (GATHER) {
            // ---- the pair sums of this camera's pairs, WAVE by wave: a wave's 64 Gaussians are 64 consecutive pairs, i.e. one
            // contiguous slot range of its own -- lane = slot through the wave's LDS rows (coalesced, the next 64 slots in
            // flight), then lane = pair adds the rows of its own slots in slot order.  No workgroup barrier: the four waves of
            // a workgroup walk their ranges independently (with 256-slot chunks behind __syncthreads the kernel took 0.31 ms
            // at four waves per SIMD; the sums and their order are the same).
            // How the rows of a chunk are added depends on how many slots a pair has (decided per wave and camera, uniform):
            //   narrow (<= 6 slots per pair on average: SYNTH-1M has 3.3)   lane = pair adds the nine values of its own rows,
            //            sums in registers -- a chunk of 64 slots meets ~20 pairs, four or five rows each;
            //   wide   (round 6; the configs[1] example after a few hundred iterations: 25 slots per pair)   the additions are
            //            dealt by (pair, value) ITEM: the pairs that meet the chunk times the nine values are spread over the
            //            lanes, each item adds its rows in slot order onto the pair's running sum in LDS.  With lane = pair
            //            such a scene left 3 of 64 lanes working through up to 64 rows of nine values each: the projection
            //            backward took 7.05 of a 13.7 ms iteration, 3.9 ms in this form.
            // Either way a pair's rows are added in slot order starting from zero: the same sums, bit for bit.
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            float* wval = sVal + wv * (64 * ROW);
            // (a thread past g_end sits on the last pair: its range is the empty one BEHIND that pair, so that lane 63 still
            // closes the wave's range)
            const int my_end = cum[pid];
            const int my_start = !valid ? my_end : (pid == 0 ? 0 : cum[pid - 1]);
            const int s0 = __builtin_amdgcn_readfirstlane(my_start);
            const int s1 = __builtin_amdgcn_readlane(my_end, 63);
            float acc[ACC_VALS];
            float2 q0, q1, q2, q3, q4;
            auto fetch = [&](int u) {
                q0 = q1 = q2 = q3 = q4 = make_float2(0.f, 0.f);   // stamp 0 = never written
                if (u < s1 && (unsigned)u < vt_cap) {   // (vt_cap: see the flush of k_blend_bwd)
                    const float2* src = reinterpret_cast<const float2*>(vtile + (int64_t)u * VT_STRIDE);
                    q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; q4 = src[4];
                }
            };
            auto rows_to_lds = [&]() {
                const bool live = __float_as_int(q4.y) == stamp;
                float* row = wval + lane * ROW;
                row[0] = live ? q0.x : 0.f; row[1] = live ? q0.y : 0.f; row[2] = live ? q1.x : 0.f;
                row[3] = live ? q1.y : 0.f; row[4] = live ? q2.x : 0.f; row[5] = live ? q2.y : 0.f;
                row[6] = live ? q3.x : 0.f; row[7] = live ? q3.y : 0.f; row[8] = live ? q4.x : 0.f;
            };
            if (s1 - s0 <= 64 * GATHER_WIDE_ABOVE) {
                if (s1 > s0) fetch(s0 + lane);
#pragma unroll
                for (int k2 = 0; k2 < ACC_VALS; ++k2) acc[k2] = 0.f;
                for (int base = s0; base < s1; base += 64) {
                    rows_to_lds();
                    wave_lds_sync();
                    fetch(base + 64 + lane);
                    const int lo = max(my_start, base) - base, hi = min(my_end, base + 64) - base;
                    for (int r = lo; r < hi; ++r) {
                        const float* src = wval + r * ROW;
#pragma unroll
                        for (int k2 = 0; k2 < ACC_VALS; ++k2) acc[k2] += src[k2];
                    }
                    wave_lds_sync();
                }
            } else {
                float* wacc = sAcc + wv * (64 * ROW);
                int* wst = sStart + wv * 65;
                wst[lane] = my_start;
                if (lane == 63) wst[64] = my_end;
#pragma unroll
                for (int k2 = 0; k2 < ACC_VALS; ++k2) wacc[lane * ROW + k2] = 0.f;
                // pairs some (record, tile) of which contributed in this backward call (k_blend_bwd marks them when the scene
                // has many slots per pair; without the marks every pair counts): a chunk of slots that meets none of them
                // holds no stamped slot -- it is neither fetched nor added
                const uint64_t tmask = touch ? __builtin_amdgcn_ballot_w64(valid && touch[pid] == (uint32_t)stamp) : ~0ull;
                auto meets = [&](int b) {
                    return __builtin_amdgcn_ballot_w64(my_end > my_start && my_start < b + 64 && my_end > b);
                };
                uint64_t pm_next = meets(s0);
                bool have = false;      // the registers hold the rows of the chunk about to be processed
                if (pm_next & tmask) { fetch(s0 + lane); have = true; }
                for (int base = s0; base < s1; base += 64) {
                    const uint64_t pm = pm_next & tmask;
                    const bool cur = have;
                    if (cur) { rows_to_lds(); wave_lds_sync(); }
                    pm_next = base + 64 < s1 ? meets(base + 64) : 0ull;
                    have = (pm_next & tmask) != 0;
                    if (have) fetch(base + 64 + lane);
                    if (cur) {
                        const int pa = __builtin_ctzll(pm), pb = 63 - __builtin_clzll(pm);
                        const int items = (pb - pa + 1) * ROW;
                        for (int it = lane; it < items; it += 64) {
                            const int q = (it * 7282) >> 16;          // it / 9 for it < 576
                            const int k2 = it - q * ROW, pp = pa + q;
                            const int lo = max(wst[pp], base) - base, hi = min(wst[pp + 1], base + 64) - base;
                            if (hi > lo) {
                                float sum = wacc[pp * ROW + k2];
                                const float* src = wval + k2;
                                int r = lo;
                                for (; r + 4 <= hi; r += 4) {   // (four reads in flight, the additions in slot order)
                                    const float a0 = src[r * ROW], a1 = src[(r + 1) * ROW], a2 = src[(r + 2) * ROW], a3 = src[(r + 3) * ROW];
                                    sum += a0; sum += a1; sum += a2; sum += a3;
                                }
                                for (; r < hi; ++r) sum += src[r * ROW];
                                wacc[pp * ROW + k2] = sum;
                            }
                        }
                        wave_lds_sync();
                    }
                }
                wave_lds_sync();   // (the zeroed sums of a range without any touched chunk are visible to their lanes)
#pragma unroll
                for (int k2 = 0; k2 < ACC_VALS; ++k2) acc[k2] = wacc[lane * ROW + k2];
                wave_lds_sync();   // (the next camera zeroes the rows)
            }
            g0 = make_float4(acc[0], acc[1], acc[2], acc[3]);
            g1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
            g2 = make_float4(acc[8], 0.f, 0.f, 0.f);
            if (!valid) continue;
        }
        const float4 r2 = splats[pid * 3 + 2];
        if (__float_as_int(r2.z) <= 0) continue;  // culled pair
        const float4 r0 = splats[pid * 3 + 0];
        const float4 r1 = splats[pid * 3 + 1];
        if (!GATHER) {
            g0 = v_splats[pid * 3 + 0];
            g1 = v_splats[pid * 3 + 1];
            g2 = v_splats[pid * 3 + 2];
        }
        const float* o = cam + c * CAM_STRIDE;
        const float R[9] = {o[0], o[1], o[2], o[4], o[5], o[6], o[8], o[9], o[10]};
        v_opac += g0.z;
        // ---- SH backward ----
        {
            float dx = mx - o[20], dy = my - o[21], dz = mz - o[22];
            float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
            float inrm = 1.0f / nrm;
            float ux = dx * inrm, uy = dy * inrm, uz = dz * inrm;
            const float vcol[3] = {g1.z, g1.w, g2.x};
            const float colf[3] = {r1.z, r1.w, r2.x};
            float vdx = 0.f, vdy = 0.f, vdz = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                // clamp_min(c+0.5, 0) passes gradient where the pre-clamp value is >= 0;
                // a clamped colour is stored as exactly 0 and a positive value is not clamped
                float pre = SH_C0 * k[ch] + SH_C1 * ((-uy * k[3 + ch] + uz * k[6 + ch]) - ux * k[9 + ch]) + 0.5f;
                float vc = (colf[ch] > 0.0f || pre >= 0.0f) ? vcol[ch] : 0.0f;
                v_k[ch] += SH_C0 * vc;
                v_k[3 + ch] += -SH_C1 * uy * vc;
                v_k[6 + ch] += SH_C1 * uz * vc;
                v_k[9 + ch] += -SH_C1 * ux * vc;
                vdx += -SH_C1 * k[9 + ch] * vc;
                vdy += -SH_C1 * k[3 + ch] * vc;
                vdz += SH_C1 * k[6 + ch] * vc;
            }
            float dotp = vdx * ux + vdy * uy + vdz * uz;
            v_mean[0] += (vdx - dotp * ux) * inrm;
            v_mean[1] += (vdy - dotp * uy) * inrm;
            v_mean[2] += (vdz - dotp * uz) * inrm;
        }
        // ---- projection backward ----
        const float x = R[0] * mx + R[1] * my + R[2] * mz + o[3];
        const float y = R[3] * mx + R[4] * my + R[5] * mz + o[7];
        const float z = R[6] * mx + R[7] * my + R[8] * mz + o[11];
        // camera covariance S = R cov R^T (symmetric, 6 unique)
        float T[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T[i * 3 + 0] = R[i * 3] * cov[0] + R[i * 3 + 1] * cov[1] + R[i * 3 + 2] * cov[2];
            T[i * 3 + 1] = R[i * 3] * cov[1] + R[i * 3 + 1] * cov[3] + R[i * 3 + 2] * cov[4];
            T[i * 3 + 2] = R[i * 3] * cov[2] + R[i * 3 + 1] * cov[4] + R[i * 3 + 2] * cov[5];
        }
        const float S00 = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
        const float S01 = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
        const float S02 = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
        const float S11 = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
        const float S12 = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
        const float S22 = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
        const float fx = o[12], fy = o[13];
        const float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const bool x_in = (xr <= o[16]) && (xr >= -o[17]);
        const bool y_in = (yr <= o[18]) && (yr >= -o[19]);
        const float tx = z * fminf(o[16], fmaxf(-o[17], xr));
        const float ty = z * fminf(o[18], fmaxf(-o[19], yr));
        const float a = fx * rz, cj = -fx * tx * rz2, b = fy * rz, d = -fy * ty * rz2;
        // conic -> cov2d
        const float A = r0.w, B = r1.x, Cc = r1.y;
        const float vA = g0.w, vB = 0.5f * g1.x, vC = g1.y;
        const float X00 = A * vA + B * vB, X01 = A * vB + B * vC;
        const float X10 = B * vA + Cc * vB, X11 = B * vB + Cc * vC;
        const float G00 = -(X00 * A + X01 * B);
        const float G01 = -0.5f * ((X00 * B + X01 * Cc) + (X10 * A + X11 * B));
        const float G11 = -(X10 * B + X11 * Cc);
        // GJ = G * J, J = [[a,0,cj],[0,b,d]]
        const float GJ00 = G00 * a, GJ01 = G01 * b, GJ02 = G00 * cj + G01 * d;
        const float GJ10 = G01 * a, GJ11 = G11 * b, GJ12 = G01 * cj + G11 * d;
        // v_S = J^T G J
        const float vS00 = a * GJ00, vS01 = a * GJ01, vS02 = a * GJ02;
        const float vS11 = b * GJ11, vS12 = b * GJ12, vS22 = cj * GJ02 + d * GJ12;
        // v_J = 2 G J S
        const float vJ00 = 2.0f * (GJ00 * S00 + GJ01 * S01 + GJ02 * S02);
        const float vJ02 = 2.0f * (GJ00 * S02 + GJ01 * S12 + GJ02 * S22);
        const float vJ11 = 2.0f * (GJ10 * S01 + GJ11 * S11 + GJ12 * S12);
        const float vJ12 = 2.0f * (GJ10 * S02 + GJ11 * S12 + GJ12 * S22);
        const float vm2x = g0.x, vm2y = g0.y;
        float vpx = fx * rz * vm2x;
        float vpy = fy * rz * vm2y;
        float vpz = -(fx * x * vm2x + fy * y * vm2y) * rz2;
        vpz += -fx * rz2 * vJ00 - fy * rz2 * vJ11;
        if (x_in) { vpx += -fx * rz2 * vJ02; vpz += 2.0f * fx * x * rz3 * vJ02; }
        else { vpz += fx * tx * rz3 * vJ02; }
        if (y_in) { vpy += -fy * rz2 * vJ12; vpz += 2.0f * fy * y * rz3 * vJ12; }
        else { vpz += fy * ty * rz3 * vJ12; }
        v_mean[0] += R[0] * vpx + R[3] * vpy + R[6] * vpz;
        v_mean[1] += R[1] * vpx + R[4] * vpy + R[7] * vpz;
        v_mean[2] += R[2] * vpx + R[5] * vpy + R[8] * vpz;
        // v_Sw += R^T vS R   (U = vS * R, then R^T * U; keep the 6 unique entries)
        float U[9];
        const float vSm[9] = {vS00, vS01, vS02, vS01, vS11, vS12, vS02, vS12, vS22};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                U[i * 3 + j] = vSm[i * 3] * R[j] + vSm[i * 3 + 1] * R[3 + j] + vSm[i * 3 + 2] * R[6 + j];
        vSw[0] += R[0] * U[0] + R[3] * U[3] + R[6] * U[6];
        vSw[1] += R[0] * U[1] + R[3] * U[4] + R[6] * U[7];
        vSw[2] += R[0] * U[2] + R[3] * U[5] + R[6] * U[8];
        vSw[3] += R[1] * U[1] + R[4] * U[4] + R[7] * U[7];
        vSw[4] += R[1] * U[2] + R[4] * U[5] + R[7] * U[8];
        vSw[5] += R[2] * U[2] + R[5] * U[5] + R[8] * U[8];
    }

    if (GATHER && !valid) return;   // (past the last barrier)
    // ---- covariance -> (quat, scale), once per Gaussian ----
    // S_w = M M^T, M = Rq diag(s):  v_M = 2 vSw M
    float Rq[9], M[9];
    asm volatile("" : "+v"(qw), "+v"(qx), "+v"(qy), "+v"(qz));   // (keeps hipcc from carrying the first evaluation across the loop)
    rot_and_m(Rq, M);
    const float W9[9] = {vSw[0], vSw[1], vSw[2], vSw[1], vSw[3], vSw[4], vSw[2], vSw[4], vSw[5]};
    float vM[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            vM[r * 3 + s] = 2.0f * (W9[r * 3] * M[s] + W9[r * 3 + 1] * M[3 + s] + W9[r * 3 + 2] * M[6 + s]);
    float v_scale[3], vR[9];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        v_scale[s] = Rq[s] * vM[s] + Rq[3 + s] * vM[3 + s] + Rq[6 + s] * vM[6 + s];
        vR[s] = vM[s] * sc[s]; vR[3 + s] = vM[3 + s] * sc[s]; vR[6 + s] = vM[6 + s] * sc[s];
    }
    // vR[i*3+j] = d/dRq[i][j]
    float vq0 = 2.0f * (qx * (vR[7] - vR[5]) + qy * (vR[2] - vR[6]) + qz * (vR[3] - vR[1]));
    float vq1 = 2.0f * (-2.0f * qx * (vR[4] + vR[8]) + qy * (vR[3] + vR[1]) + qz * (vR[6] + vR[2]) + qw * (vR[7] - vR[5]));
    float vq2 = 2.0f * (qx * (vR[3] + vR[1]) - 2.0f * qy * (vR[0] + vR[8]) + qz * (vR[7] + vR[5]) + qw * (vR[2] - vR[6]));
    float vq3 = 2.0f * (qx * (vR[6] + vR[2]) + qy * (vR[7] + vR[5]) - 2.0f * qz * (vR[0] + vR[4]) + qw * (vR[3] - vR[1]));
    const float dq = vq0 * qw + vq1 * qx + vq2 * qy + vq3 * qz;
    vq0 = (vq0 - dq * qw) * inv_norm; vq1 = (vq1 - dq * qx) * inv_norm;
    vq2 = (vq2 - dq * qy) * inv_norm; vq3 = (vq3 - dq * qz) * inv_norm;

    // ---- regularisers: k_o * d sigmoid(o), k_s * d exp(s) ----
    const float sg = 1.0f / (1.0f + __expf(-opac));
    v_opac += reg_o_k * sg * (1.0f - sg);
    v_scale[0] += reg_s_k * __expf(s0);
    v_scale[1] += reg_s_k * __expf(s1);
    v_scale[2] += reg_s_k * __expf(s2);

    // Block layout means[3 Nl] quats[4 Nl] scales[3 Nl] opacities[Nl] sh4[12 Nl] -- over all N Gaussians (the caller's
    // gradient buffer), or, range_major (the range-wise exchange of comm.hip), over the Gaussians [g_begin, g_end) of this
    // launch only, stored at 23 * g_begin: every range is then ONE contiguous piece of 23 * (g_end - g_begin) floats,
    // i.e. one all-reduce per range instead of five.
    const int64_t Nl = range_major ? g_end - g_begin : N;
    float* gb = range_major ? grads + (int64_t)23 * g_begin : grads;
    const int gw = range_major ? g - g_begin : g;   // parameters were read at g, gradients are written at gw
    float* gm = gb; float* gq = gb + 3 * Nl; float* gs = gb + 7 * Nl;
    float* go = gb + 10 * Nl; float* gsh = gb + 11 * Nl;
    if (accumulate) {   // a later view chunk of the same training call: its gradients add to the earlier chunks
        v_mean[0] += gm[3 * gw]; v_mean[1] += gm[3 * gw + 1]; v_mean[2] += gm[3 * gw + 2];
        vq0 += gq[4 * gw]; vq1 += gq[4 * gw + 1]; vq2 += gq[4 * gw + 2]; vq3 += gq[4 * gw + 3];
        v_scale[0] += gs[3 * gw]; v_scale[1] += gs[3 * gw + 1]; v_scale[2] += gs[3 * gw + 2];
        v_opac += go[gw];
#pragma unroll
        for (int i = 0; i < 12; ++i) v_k[i] += gsh[(int64_t)gw * 12 + i];
    }
    gm[3 * gw] = v_mean[0]; gm[3 * gw + 1] = v_mean[1]; gm[3 * gw + 2] = v_mean[2];
    gq[4 * gw] = vq0; gq[4 * gw + 1] = vq1; gq[4 * gw + 2] = vq2; gq[4 * gw + 3] = vq3;
    gs[3 * gw] = v_scale[0]; gs[3 * gw + 1] = v_scale[1]; gs[3 * gw + 2] = v_scale[2];
    go[gw] = v_opac;
#pragma unroll
    for (int i = 0; i < 12; ++i) gsh[(int64_t)gw * 12 + i] = v_k[i];
}

int st3r_project_sh_bwd_impl(hipStream_t s, int N, int C, const float* means, const float* quats, const float* scales,
                             const float* opacities, const float* sh, int sh_stride, const float* viewmats,
                             const float* Ks, const float* campos, int width, int height, float eps2d,
                             const float* splats, const float* v_splats, float reg_views, float opac_fac,
                             float scale_fac, float* grads, bool accumulate, int g_begin, int g_end, bool range_major,
                             const st3r_vtile_ref* slots) {
    if (g_end < 0) g_end = N;
    if (N == 0 || g_end <= g_begin) return ST3R_OK;
    float reg_o_k = reg_views * opac_fac / (float)N;
    float reg_s_k = reg_views * scale_fac / (3.0f * (float)N);
    size_t shmem = (size_t)C * CAM_STRIDE * sizeof(float);
    if (slots)   // fused training calls: the pair sums are gathered from the backward's slots in this kernel
        hipLaunchKernelGGL(k_project_sh_bwd<true>, dim3(ceil_div(g_end - g_begin, 256)), dim3(256), shmem, s, N, C, means,
                           quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, width, height, eps2d,
                           (const float4*)splats, (const float4*)nullptr, reg_o_k, reg_s_k, grads, accumulate ? 1 : 0, g_begin,
                           g_end, range_major ? 1 : 0, slots->cum, slots->vtile, slots->stamp, slots->vt_cap, slots->touch);
    else
        hipLaunchKernelGGL(k_project_sh_bwd<false>, dim3(ceil_div(g_end - g_begin, 256)), dim3(256), shmem, s, N, C, means,
                           quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, width, height, eps2d,
                           (const float4*)splats, (const float4*)v_splats, reg_o_k, reg_s_k, grads, accumulate ? 1 : 0, g_begin,
                           g_end, range_major ? 1 : 0, (const int32_t*)nullptr, (const float*)nullptr, 0, 0u, (const uint32_t*)nullptr);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_project_sh_bwd(st3r_ctx* ctx, void* stream, int N, int C, const float* means,
                                       const float* quats, const float* scales, const float* opacities,
                                       const float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                       const float* campos, int width, int height, float eps2d,
                                       const float* splats, const float* v_splats, float reg_views,
                                       float opac_fac, float scale_fac, float* grads) {
    ARG_CHECK(ctx && N >= 0 && C > 0 && C <= ST3R_MAX_VIEWS && sh_stride >= 12);
    ARG_CHECK(means && quats && scales && opacities && sh && viewmats && Ks && campos && splats && v_splats && grads);
    return st3r_project_sh_bwd_impl((hipStream_t)stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats,
                                    Ks, campos, width, height, eps2d, splats, v_splats, reg_views, opac_fac, scale_fac,
                                    grads, false, 0, -1, false, nullptr);
}
