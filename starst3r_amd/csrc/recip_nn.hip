// K10: nearest neighbour under the dot-product similarity, the inner operation of Mast3r's
// fast_reciprocal_NNs (reached from starster/reconstruct.py:97 forward_mast3r -> extract_correspondences ->
// fast_reciprocal_NNs(dist='dot', block_size=2**13)):
//     nn[q] = argmax_j  Q[q] . DB[j]        (first index on ties)
// for n queries (<= 3072 seeds, shrinking every iteration) against the m = H*W descriptors (D = 24)
// of the other image.
//
// MFMA block-matmul with a fused running arg-max -- the n x m score matrix is never written.
// v_mfma_f32_32x32x2_f32 (exact fp32: each product rounded once, fp32 accumulate) with the operands
// "swapped": DB rows are the M dimension, queries the N dimension, so lane l holds, for ITS query
// column (l & 31), 16 different DB rows per tile.  The arg-max over DB rows is therefore a per-lane
// register scan (compare + two selects per row), no cross-lane traffic until the two half-waves that
// share a column are merged once at the very end.
//   K = 24 is split by half-wave: lanes 0-31 feed components 0..11, lanes 32-63 components 12..23, so every
//   lane reads 48 contiguous bytes of its DB row (3 x dwordx4) and 12 MFMAs cover the whole dot product.
//   A wave owns 64 queries (two 32-column tiles that reuse the same DB fragment) and walks one of S
//   segments of the DB; per-(query, segment) winners are reduced by a second tiny kernel.
// The DB (18.9 MB at 512x384) is read from HBM once per call and re-used out of L2 / Infinity Cache by
// the other query groups.  Roofline: MFMA fp32 (157 TFLOP/s), useful flops 2*n*m*24.
#include "common.h"

#define NN_D 24
#define NN_HALF 12

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Query i reads row qsel[act[i]] of Q when the indirections are given (device-resident reciprocal loop: `act`
// lists the seeds that have not converged, `n_dev` holds how many there are), row i otherwise.
__global__ __launch_bounds__(256) void k_nn_argmax(const float* __restrict__ Q, int n, const float* __restrict__ DB,
                                                   int m, int S, int tiles_per_seg, float* __restrict__ part_val,
                                                   int32_t* __restrict__ part_idx, const int32_t* __restrict__ qsel,
                                                   const int32_t* __restrict__ act, const int32_t* __restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = (n + 63) >> 6;
    if (wid >= groups * S) return;
    const int group = wid / S, seg = wid - group * S;
    if (n_dev) { n = *n_dev; if (group * 64 >= n) return; }
    const int j = lane & 31, h = lane >> 5;
    // query fragments: column j of tile t, components 12h .. 12h+11
    float q[2][NN_HALF];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = group * 64 + t * 32 + j;
        if (qi < n) {
            const int slot = act ? act[qi] : qi;
            const int64_t qrow = qsel ? qsel[slot] : slot;
            const float4* src = reinterpret_cast<const float4*>(Q + qrow * NN_D + NN_HALF * h);
            const float4 a = src[0], b = src[1], c = src[2];
            q[t][0] = a.x; q[t][1] = a.y; q[t][2] = a.z; q[t][3] = a.w; q[t][4] = b.x; q[t][5] = b.y;
            q[t][6] = b.z; q[t][7] = b.w; q[t][8] = c.x; q[t][9] = c.y; q[t][10] = c.z; q[t][11] = c.w;
        } else {
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) q[t][k] = 0.f;
        }
    }
    // running winner per query tile: value, the tile it came from and the row's position r in the lane's 16 scores
    // (the DB row is tile * 32 + (r & 3) + 8 (r >> 2) + 4 h, rebuilt once at the end) -- per score one compare and two
    // selects with inline constants; the tile is noted once per tile and only the last tile checks rows against m
    float best[2] = {-INFINITY, -INFINITY};
    int btile[2] = {-1, -1}, bcode[2] = {0, 0};
    const int tile0 = seg * tiles_per_seg;
    const int tile1 = min(tile0 + tiles_per_seg, (m + 31) >> 5);
    const int full1 = min(tile1, m >> 5);   // tiles [tile0, full1) lie entirely below m
    // the lane's 48 bytes of the next tile travel while the 24 MFMAs of the current one run
    float4 nx = make_float4(0, 0, 0, 0), ny = nx, nz = nx;
    if (tile0 < full1) {
        const float4* src = reinterpret_cast<const float4*>(DB + (int64_t)(tile0 * 32 + j) * NN_D + NN_HALF * h);
        nx = src[0]; ny = src[1]; nz = src[2];
    }
    for (int tile = tile0; tile < full1; ++tile) {
        const float4 x = nx, y = ny, z = nz;
        if (tile + 1 < full1) {
            const float4* src = reinterpret_cast<const float4*>(DB + (int64_t)((tile + 1) * 32 + j) * NN_D + NN_HALF * h);
            nx = src[0]; ny = src[1]; nz = src[2];
        }
        const float a[NN_HALF] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y, z.z, z.w};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], q[t][k], c, 0, 0, 0);
            const float before = best[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = c[r];
                const bool better = v > best[t];  // strict: the first (smallest) index wins ties (rows increase with r)
                best[t] = better ? v : best[t];
                bcode[t] = better ? r : bcode[t];
            }
            btile[t] = best[t] > before ? tile : btile[t];
        }
    }
    int bidx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        bidx[t] = btile[t] < 0 ? 0x7fffffff : btile[t] * 32 + (bcode[t] & 3) + 8 * (bcode[t] >> 2) + 4 * h;
    for (int tile = full1; tile < tile1; ++tile) {   // the ragged last tile of the DB
        const int row = tile * 32 + j;  // this lane's DB row for the A operand
        float a[NN_HALF];
        if (row < m) {
            const float4* src = reinterpret_cast<const float4*>(DB + (int64_t)row * NN_D + NN_HALF * h);
            const float4 x = src[0], y = src[1], z = src[2];
            a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w; a[4] = y.x; a[5] = y.y; a[6] = y.z; a[7] = y.w;
            a[8] = z.x; a[9] = z.y; a[10] = z.z; a[11] = z.w;
        } else {
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) a[k] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], q[t][k], c, 0, 0, 0);
            // lane holds scores of DB rows (r&3) + 8(r>>2) + 4h of this tile for query column j: increasing in r
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = c[r];
                const bool better = (rr < m) && (v > best[t]);  // strict: the first (smallest) index wins ties
                best[t] = better ? v : best[t];
                bidx[t] = better ? rr : bidx[t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float ov = __shfl_xor(best[t], 32);
        const int oi = __shfl_xor(bidx[t], 32);
        if (ov > best[t] || (ov == best[t] && oi < bidx[t])) { best[t] = ov; bidx[t] = oi; }
        const int qi = group * 64 + t * 32 + j;
        if (h == 0 && qi < n) {
            part_val[(int64_t)qi * S + seg] = best[t];
            part_idx[(int64_t)qi * S + seg] = bidx[t];
        }
    }
}

// winner over the S segment partials of one query, one wave per query (S can be ~1000 when few queries are
// active: a serial scan per thread would dominate the call)
__device__ __forceinline__ void wave_best(const float* __restrict__ pv, const int32_t* __restrict__ pi, int S,
                                          float* bv_out, int* bi_out) {
    const int lane = threadIdx.x & 63;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
        const float v = pv[s];
        const int i = pi[s];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    *bv_out = bv; *bi_out = bi;
}

__global__ __launch_bounds__(256) void k_nn_reduce(int n, int S, const float* __restrict__ part_val,
                                                   const int32_t* __restrict__ part_idx, int32_t* __restrict__ nn,
                                                   float* __restrict__ score) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n) return;
    float bv; int bi;
    wave_best(part_val + (int64_t)qi * S, part_idx + (int64_t)qi * S, S, &bv, &bi);
    if ((threadIdx.x & 63) == 0) {
        nn[qi] = bi;
        if (score) score[qi] = bv;
    }
}

ST3R_EXPORT int st3r_nn_dot_argmax(st3r_ctx* ctx, void* stream, const float* queries, int n, const float* db,
                                   int m, int dim, int32_t* nn_out, float* score_out) {
    ARG_CHECK(ctx && n >= 0 && m > 0 && dim == NN_D && (n == 0 || (queries && nn_out)) && db);
    if (n == 0) return ST3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int groups = (n + 63) / 64;
    const int tiles = (m + 31) / 32;
    int S = 4096 / groups;           // enough waves to fill 256 CUs a few times over
    if (S < 1) S = 1;
    if (S > (tiles + 3) / 4) S = (tiles + 3) / 4;  // at least ~4 tiles per segment
    if (S < 1) S = 1;
    const int tiles_per_seg = (tiles + S - 1) / S;
    S = (tiles + tiles_per_seg - 1) / tiles_per_seg;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_NN_PART, (sizeof(float) + sizeof(int32_t)) * (size_t)n * S, &p);
    if (rc) return rc;
    float* part_val = (float*)p;
    int32_t* part_idx = (int32_t*)(part_val + (size_t)n * S);
    const int waves = groups * S;
    hipLaunchKernelGGL(k_nn_argmax, dim3((waves + 3) / 4), dim3(256), 0, s, queries, n, db, m, S, tiles_per_seg,
                       part_val, part_idx, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_nn_reduce, dim3(ceil_div(n, 4)), dim3(256), 0, s, n, S, part_val, part_idx, nn_out, score_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}

// ---- device-resident reciprocal loop (Mast3r fast_reciprocal_NNs) ----

// seeds on the grid S/2, S/2+S, ... in x and y (np.mgrid[S//2:H:S, S//2:W:S], flat index x + W*y, ascending)
__global__ void k_nn_seeds(int nx, int ny, int S, int W, int32_t* __restrict__ xy1, int32_t* __restrict__ xy2,
                           int32_t* __restrict__ notyet) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = S / 2 + (i / nx) * S, x = S / 2 + (i % nx) * S;
    xy1[i] = x + W * y; xy2[i] = -1; notyet[i] = 1;
}

// act[] = indices with notyet != 0, in ascending order; one block, chunked LDS scan
__global__ __launch_bounds__(1024) void k_nn_compact(int n, const int32_t* __restrict__ notyet,
                                                     int32_t* __restrict__ act, int32_t* __restrict__ n_act) {
    __shared__ int sWave[16];
    __shared__ int sBase;
    if (threadIdx.x == 0) sBase = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const bool on = i < n && notyet[i] != 0;
        const uint64_t ball = __ballot(on);
        const int before = __popcll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) sWave[w] = __popcll(ball);
        __syncthreads();
        int off = sBase;
        for (int k = 0; k < w; ++k) off += sWave[k];
        if (on) act[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += sWave[k]; sBase += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_act = sBase;
}

// winner over the segments of active query i -> dst[act[i]]; a seed whose neighbour did not change has converged
__global__ __launch_bounds__(256) void k_nn_reduce_update(const int32_t* __restrict__ n_act, int S,
                                                          const float* __restrict__ part_val,
                                                          const int32_t* __restrict__ part_idx,
                                                          const int32_t* __restrict__ act, int32_t* __restrict__ dst,
                                                          int32_t* __restrict__ notyet) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= *n_act) return;
    float bv; int bi;
    wave_best(part_val + (int64_t)i * S, part_idx + (int64_t)i * S, S, &bv, &bi);
    if ((threadIdx.x & 63) == 0) {
        const int slot = act[i];
        if (dst[slot] == bi) notyet[slot] = 0;  // notyet &= (old != new)
        dst[slot] = bi;
    }
}

ST3R_EXPORT int st3r_recip_nn_seed_count(int H1, int W1, int subsample) {
    if (H1 <= 0 || W1 <= 0 || subsample <= 0) return 0;
    const int S = subsample;
    const int ny = (H1 - S / 2 + S - 1) / S, nx = (W1 - S / 2 + S - 1) / S;
    return (ny > 0 && nx > 0) ? nx * ny : 0;
}

ST3R_EXPORT int st3r_recip_nn(st3r_ctx* ctx, void* stream, const float* descA, int H1, int W1, const float* descB,
                              int H2, int W2, int dim, int subsample, int max_iter, int32_t* idx1_out,
                              int32_t* idx2_out, int32_t* notyet_out) {
    ARG_CHECK(ctx && descA && descB && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && dim == NN_D && subsample > 0);
    ARG_CHECK(max_iter > 0 && idx1_out && idx2_out && notyet_out);
    const int S0 = subsample;
    const int ny = (H1 - S0 / 2 + S0 - 1) / S0, nx = (W1 - S0 / 2 + S0 - 1) / S0;
    const int n = nx * ny;
    if (n <= 0) return ST3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int mA = H1 * W1, mB = H2 * W2;
    const int groups = (n + 63) / 64;
    auto plan = [&](int m, int* S, int* tps) {
        const int tiles = (m + 31) / 32;
        int Sx = 4096 / groups;
        if (Sx < 1) Sx = 1;
        if (Sx > (tiles + 3) / 4) Sx = (tiles + 3) / 4;
        if (Sx < 1) Sx = 1;
        *tps = (tiles + Sx - 1) / Sx;
        *S = (tiles + *tps - 1) / *tps;
    };
    int SA, tpsA, SB, tpsB;
    plan(mA, &SA, &tpsA); plan(mB, &SB, &tpsB);
    const int Smax = SA > SB ? SA : SB;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_NN_PART, (sizeof(float) + sizeof(int32_t)) * (size_t)n * Smax +
                                                   sizeof(int32_t) * ((size_t)n + 4), &p);
    if (rc) return rc;
    float* part_val = (float*)p;
    int32_t* part_idx = (int32_t*)(part_val + (size_t)n * Smax);
    int32_t* act = part_idx + (size_t)n * Smax;
    int32_t* n_act = act + n;
    hipLaunchKernelGGL(k_nn_seeds, dim3(ceil_div(n, 256)), dim3(256), 0, s, nx, ny, S0, W1, idx1_out, idx2_out, notyet_out);
    for (int it = 0; it < max_iter; ++it) {
        // xy2 = NN_B(A[xy1]) for the seeds still moving
        hipLaunchKernelGGL(k_nn_compact, dim3(1), dim3(1024), 0, s, n, notyet_out, act, n_act);
        hipLaunchKernelGGL(k_nn_argmax, dim3((groups * SB + 3) / 4), dim3(256), 0, s, descA, n, descB, mB, SB, tpsB,
                           part_val, part_idx, (const int32_t*)idx1_out, (const int32_t*)act, (const int32_t*)n_act);
        hipLaunchKernelGGL(k_nn_reduce_update, dim3(ceil_div(n, 4)), dim3(256), 0, s, n_act, SB, part_val, part_idx, act,
                           idx2_out, notyet_out);
        // xy1 = NN_A(B[xy2])
        hipLaunchKernelGGL(k_nn_compact, dim3(1), dim3(1024), 0, s, n, notyet_out, act, n_act);
        hipLaunchKernelGGL(k_nn_argmax, dim3((groups * SA + 3) / 4), dim3(256), 0, s, descB, n, descA, mA, SA, tpsA,
                           part_val, part_idx, (const int32_t*)idx2_out, (const int32_t*)act, (const int32_t*)n_act);
        hipLaunchKernelGGL(k_nn_reduce_update, dim3(ceil_div(n, 4)), dim3(256), 0, s, n_act, SA, part_val, part_idx, act,
                           idx1_out, notyet_out);
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}
