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

__global__ __launch_bounds__(256) void k_nn_argmax(const float* __restrict__ Q, int n, const float* __restrict__ DB,
                                                   int m, int S, int tiles_per_seg, float* __restrict__ part_val,
                                                   int32_t* __restrict__ part_idx) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = (n + 63) >> 6;
    if (wid >= groups * S) return;
    const int group = wid / S, seg = wid - group * S;
    const int j = lane & 31, h = lane >> 5;
    // query fragments: column j of tile t, components 12h .. 12h+11
    float q[2][NN_HALF];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = group * 64 + t * 32 + j;
        if (qi < n) {
            const float4* src = reinterpret_cast<const float4*>(Q + (int64_t)qi * NN_D + NN_HALF * h);
            const float4 a = src[0], b = src[1], c = src[2];
            q[t][0] = a.x; q[t][1] = a.y; q[t][2] = a.z; q[t][3] = a.w; q[t][4] = b.x; q[t][5] = b.y;
            q[t][6] = b.z; q[t][7] = b.w; q[t][8] = c.x; q[t][9] = c.y; q[t][10] = c.z; q[t][11] = c.w;
        } else {
#pragma unroll
            for (int k = 0; k < NN_HALF; ++k) q[t][k] = 0.f;
        }
    }
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {0x7fffffff, 0x7fffffff};
    const int tile0 = seg * tiles_per_seg;
    const int tile1 = min(tile0 + tiles_per_seg, (m + 31) >> 5);
    for (int tile = tile0; tile < tile1; ++tile) {
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

__global__ __launch_bounds__(256) void k_nn_reduce(int n, int S, const float* __restrict__ part_val,
                                                   const int32_t* __restrict__ part_idx, int32_t* __restrict__ nn,
                                                   float* __restrict__ score) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n) return;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int s = 0; s < S; ++s) {
        const float v = part_val[(int64_t)qi * S + s];
        const int i = part_idx[(int64_t)qi * S + s];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    nn[qi] = bi;
    if (score) score[qi] = bv;
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
                       part_val, part_idx);
    hipLaunchKernelGGL(k_nn_reduce, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, S, part_val, part_idx, nn_out, score_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}
