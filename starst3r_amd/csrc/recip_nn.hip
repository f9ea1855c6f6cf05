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
// register operation (a v_max3 tree per tile; the row inside the winning tile is found once, after the walk), no
// cross-lane traffic until the two half-waves that share a column are merged once at the very end.
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

// maximum of the 16 scores a lane holds after the 12 MFMAs of a tile: a v_max3_f32 tree (inline asm: the builtin
// fmax carries IEEE canonicalisations that the scores -- finite sums of finite products -- do not need)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max16(const f32x16& c) {
    const float m0 = max3f(c[0], c[1], c[2]), m1 = max3f(c[3], c[4], c[5]), m2 = max3f(c[6], c[7], c[8]);
    const float m3 = max3f(c[9], c[10], c[11]), m4 = max3f(c[12], c[13], c[14]);
    const float m5 = max3f(m0, m1, m2), m6 = max3f(m3, m4, c[15]);
    return m5 > m6 ? m5 : m6;
}

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
    // Running winner per query tile: the best score and the TILE it came from -- per (tile, query tile) a v_max3 tree over
    // the lane's 16 scores (8 instructions), one compare and two selects.  WHICH of the tile's 16 rows it was is settled
    // once per query, by the reduction kernel (wave_resolve).  Round 4 tracked the row inside the walk with a compare and
    // two selects per score: 100 VALU instructions per 24 MFMAs.  A SIMD issues one instruction at a time and an MFMA
    // holds the slot for its 64 cycles (tools/probe/nn_walk_probe.hip: the bare walk 131-138 TFLOP/s, + this epilogue
    // -5 %, + the three loads and their addressing -8 %), so every instruction that is not an MFMA is paid in full.
    float best[2] = {-INFINITY, -INFINITY};
    int btile[2] = {-1, -1};
    const int tile0 = seg * tiles_per_seg;
    const int tile1 = min(tile0 + tiles_per_seg, (m + 31) >> 5);
    const int full1 = min(tile1, m >> 5);   // tiles [tile0, full1) lie entirely below m
    auto walk = [&](int tile, const float4& x, const float4& y, const float4& z) {
        const float a[NN_HALF] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y, z.z, z.w};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], q[t][k], c, 0, 0, 0);
            const float tm = max16(c);
            const bool better = tm > best[t];   // strict: of equal scores the earlier tile (smaller rows) stays
            best[t] = better ? tm : best[t];
            btile[t] = better ? tile : btile[t];
        }
    };
    int tile = tile0;
    const float* dbl = DB + (int64_t)j * NN_D + NN_HALF * h;   // this lane's 48 bytes of tile 0
    auto fetch = [&](int tile, float4& x, float4& y, float4& z) {
        const float4* src = reinterpret_cast<const float4*>(dbl + (int64_t)tile * (32 * NN_D));
        x = src[0]; y = src[1]; z = src[2];
    };
    // two register sets take turns: the lane's 48 bytes of the next tile travel while the 24 MFMAs of the current one run
    // (one set + a rotation cost 24 v_mov per tile)
    float4 ax, ay, az, bx, by, bz;
    ax = ay = az = bx = by = bz = make_float4(0, 0, 0, 0);
    if (tile < full1) fetch(tile, ax, ay, az);
    for (; tile + 1 < full1; tile += 2) {
        fetch(tile + 1, bx, by, bz);
        walk(tile, ax, ay, az);
        if (tile + 2 < full1) fetch(tile + 2, ax, ay, az);
        walk(tile + 1, bx, by, bz);
    }
    if (tile < full1) walk(tile, ax, ay, az);
    for (tile = full1; tile < tile1; ++tile) {   // the ragged last tile of the DB: rows >= m score -inf
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
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                tm = (rr < m && c[r] > tm) ? c[r] : tm;
            }
            const bool better = tm > best[t];
            best[t] = better ? tm : best[t];
            btile[t] = better ? tile : btile[t];
        }
    }
    // per (query, segment, half-wave): the best score and its tile.  WHICH of the half-wave's 16 rows of that tile it
    // was is settled by the reduction kernel, once per query instead of once per (query, segment, half) -- done here,
    // at the end of every wave, the second look cost 53 us of 333 (latency-bound gathers with nothing left to overlap).
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = group * 64 + t * 32 + j;
        if (qi < n) {
            const int64_t o = ((int64_t)qi * S + seg) * 2 + h;
            part_val[o] = best[t];
            part_idx[o] = btile[t];
        }
    }
}

// Winner over the 2 S (segment, half-wave) partials of one query, one wave per query (S can be ~1000 when few queries
// are active: a serial scan per thread would dominate the call).  A partial is (best score, tile) of a half-wave h: its
// 16 candidate rows are tile * 32 + 4 h + (r & 3) + 8 (r >> 2), r = 0 .. 15.  The winning partial's rows are scored
// again by the whole wave in plain fp32 -- four lanes per row, six products each, (p0 + p1) + (p2 + p3) -- and the first
// maximum in row order is the neighbour.  Identical rows score identically here as they do in the MFMA, so exact ties
// still go to the first index: EVERY partial whose score equals the maximum is looked at (usually one) and the smallest
// winning row stands.  Rows that differ by an ulp are numerical ties for any fp32 sum order (tests/test_gpu_nn.py holds
// indices to a 1e-5 relative gap between best and runner-up).
__device__ __forceinline__ void wave_resolve(const float* __restrict__ pv, const int32_t* __restrict__ pt, int S2,
                                             const float* __restrict__ qv, const float* __restrict__ DB, int m,
                                             float* bv_out, int* bi_out) {
    const int lane = threadIdx.x & 63;
    float bv = -INFINITY;
    for (int s = lane; s < S2; s += 64) {
        const float v = pv[s];
        bv = (pt[s] >= 0 && v > bv) ? v : bv;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bv = fmaxf(bv, __shfl_xor(bv, o));
    const int r = lane >> 2, part = lane & 3;
    const float2* q2 = reinterpret_cast<const float2*>(qv + 6 * part);
    const float2 qa = q2[0], qb = q2[1], qc = q2[2];
    int bi = 0x7fffffff;
    for (int s0 = 0; s0 < S2; s0 += 64) {
        const int s = s0 + lane;
        const bool tied = s < S2 && pt[s] >= 0 && pv[s] == bv;
        uint64_t todo = __ballot(tied);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int cs = s0 + src;                       // wave-uniform candidate: partial cs = segment * 2 + h
            const int tile = __shfl(tied ? pt[s] : 0, src);
            const int rr = tile * 32 + 4 * (cs & 1) + (r & 3) + 8 * (r >> 2);
            float sc = -INFINITY;
            if (rr < m) {
                const float2* d2 = reinterpret_cast<const float2*>(DB + (int64_t)rr * NN_D + 6 * part);
                const float2 da = d2[0], db = d2[1], dc = d2[2];
                sc = fmaf(dc.y, qc.y, fmaf(dc.x, qc.x, fmaf(db.y, qb.y, fmaf(db.x, qb.x, fmaf(da.y, qa.y, da.x * qa.x)))));
            }
            sc += __shfl_xor(sc, 1);       // (p0 + p1), (p2 + p3)
            sc += __shfl_xor(sc, 2);       // + : the same order in all four lanes of a row
            int ri = rr < m ? rr : 0x7fffffff;
#pragma unroll
            for (int o = 4; o <= 32; o <<= 1) {            // first maximum in row order over the 16 rows
                const float os = __shfl_xor(sc, o);
                const int oi = __shfl_xor(ri, o);
                if (os > sc || (os == sc && oi < ri)) { sc = os; ri = oi; }
            }
            bi = min(bi, ri);
        }
    }
    *bv_out = bv; *bi_out = bi;
}

__global__ __launch_bounds__(256) void k_nn_reduce(int n, int S, const float* __restrict__ part_val,
                                                   const int32_t* __restrict__ part_idx, const float* __restrict__ Q,
                                                   const float* __restrict__ DB, int m, int32_t* __restrict__ nn,
                                                   float* __restrict__ score) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n) return;
    float bv; int bi;
    wave_resolve(part_val + (int64_t)qi * S * 2, part_idx + (int64_t)qi * S * 2, 2 * S, Q + (int64_t)qi * NN_D, DB, m, &bv, &bi);
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
    int rc = st3r_arena_get(ctx, SLOT_NN_PART, (sizeof(float) + sizeof(int32_t)) * (size_t)n * S * 2, &p);
    if (rc) return rc;
    float* part_val = (float*)p;
    int32_t* part_idx = (int32_t*)(part_val + (size_t)n * S * 2);
    const int waves = groups * S;
    hipLaunchKernelGGL(k_nn_argmax, dim3((waves + 3) / 4), dim3(256), 0, s, queries, n, db, m, S, tiles_per_seg,
                       part_val, part_idx, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_nn_reduce, dim3(ceil_div(n, 4)), dim3(256), 0, s, n, S, part_val, part_idx, queries, db, m, nn_out,
                       score_out);
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
                                                          const float* __restrict__ Q, const int32_t* __restrict__ qsel,
                                                          const float* __restrict__ DB, int m,
                                                          const int32_t* __restrict__ act, int32_t* __restrict__ dst,
                                                          int32_t* __restrict__ notyet) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= *n_act) return;
    float bv; int bi;
    const int slot = act[i];
    wave_resolve(part_val + (int64_t)i * S * 2, part_idx + (int64_t)i * S * 2, 2 * S, Q + (int64_t)qsel[slot] * NN_D, DB, m,
                 &bv, &bi);
    if ((threadIdx.x & 63) == 0) {
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
    int rc = st3r_arena_get(ctx, SLOT_NN_PART, (sizeof(float) + sizeof(int32_t)) * (size_t)n * Smax * 2 +
                                                   sizeof(int32_t) * ((size_t)n + 4), &p);
    if (rc) return rc;
    float* part_val = (float*)p;
    int32_t* part_idx = (int32_t*)(part_val + (size_t)n * Smax * 2);
    int32_t* act = part_idx + (size_t)n * Smax * 2;
    int32_t* n_act = act + n;
    hipLaunchKernelGGL(k_nn_seeds, dim3(ceil_div(n, 256)), dim3(256), 0, s, nx, ny, S0, W1, idx1_out, idx2_out, notyet_out);
    for (int it = 0; it < max_iter; ++it) {
        // xy2 = NN_B(A[xy1]) for the seeds still moving
        hipLaunchKernelGGL(k_nn_compact, dim3(1), dim3(1024), 0, s, n, notyet_out, act, n_act);
        hipLaunchKernelGGL(k_nn_argmax, dim3((groups * SB + 3) / 4), dim3(256), 0, s, descA, n, descB, mB, SB, tpsB,
                           part_val, part_idx, (const int32_t*)idx1_out, (const int32_t*)act, (const int32_t*)n_act);
        hipLaunchKernelGGL(k_nn_reduce_update, dim3(ceil_div(n, 4)), dim3(256), 0, s, n_act, SB, part_val, part_idx, descA,
                           (const int32_t*)idx1_out, descB, mB, act, idx2_out, notyet_out);
        // xy1 = NN_A(B[xy2])
        hipLaunchKernelGGL(k_nn_compact, dim3(1), dim3(1024), 0, s, n, notyet_out, act, n_act);
        hipLaunchKernelGGL(k_nn_argmax, dim3((groups * SA + 3) / 4), dim3(256), 0, s, descB, n, descA, mA, SA, tpsA,
                           part_val, part_idx, (const int32_t*)idx2_out, (const int32_t*)act, (const int32_t*)n_act);
        hipLaunchKernelGGL(k_nn_reduce_update, dim3(ceil_div(n, 4)), dim3(256), 0, s, n_act, SA, part_val, part_idx, descB,
                           (const int32_t*)idx2_out, descA, mA, act, idx1_out, notyet_out);
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}
