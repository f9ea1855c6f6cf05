// K6/K7: per-tile alpha blending, forward and backward.
// Replaces gsplat rasterize_to_pixels fwd/bwd (starster/gs.py:76 and the autograd pass of
// starster/gs.py:153).
//
// Mapping.  One workgroup = one 16x16 tile = 4 wave64; wave w owns the 8x8 quadrant
// (w&1, w>>1).  Workgroups are remapped so that each XCD (private 4 MiB L2) walks groups of
// consecutive (camera, tile) ids: one tile row per group, the rows round-robin over the XCDs (see xcd_remap, XCD_GROUP).
//
// Staging + wavefront compaction.  The tile's depth-sorted list is staged through LDS in
// batches of 256 splat records (one record per thread, 3 x 16-byte loads).  The staging
// thread also tests the record's ellipse {alpha >= 1/255} (tau slightly inflated) exactly
// against the four quadrants (tile_rect.h: ellipse_hits_square); four wave ballots per
// 64 records give every wave a 256-bit "relevant" mask, and the wave then walks only the set
// bits with scalar bit-scan instructions.  Records that cannot reach a quadrant cost that
// wave nothing (in the reference they fail the alpha test on all 64 pixels).
//
// Forward -> backward hand-off.  While blending, each wave records which staged records
// actually contributed to at least one of its pixels (one bit per record per wave, ~20 MB at
// 1M Gaussians / 8x1080p) and the number of batches the tile consumed.  The backward pass
// walks exactly those bits back to front: no alpha test for non-contributors, no work behind
// a wave's last contribution.
//
// Backward, two phases per wave.  Summing nine gradients over the 64 pixels of a wave for every record costs more
// than computing them (a 64-lane butterfly per record), so the wave transposes instead:
//   phase 1 (lanes = pixels)   walks the contributing records back to front exactly like the forward pass walks
//                              them front to back, and leaves two scalars per (record, pixel) in a wave-private LDS
//                              buffer: g_o = vis * dL/dalpha and fac = alpha * T.
//   phase 2 (lanes = records)  every CHUNK records the lanes regroup as (record, pixel run): each lane walks CHUNK
//                              pixels of one row and accumulates the record's sums
//                              sum g_o {1, dx, dy, dx^2, dx dy, dy^2} and sum fac v_rgb in registers; the partial
//                              sums of a record meet in one halving butterfly per CHUNK records.
// A wave meets a record once per round, so its sums are stored (not added) into the wave's own accumulator rows;
// after the round one thread per record adds the rows of the waves whose contribution bit is set, applies the
// record constants (opacity, conic) and writes the record's stamped slot in HBM -- no atomics anywhere;
// k_gather_vtile sums the slots of a (camera, gaussian) pair in order.  Rounds stage HB = 64 records (a quarter of
// a forward batch) and CHUNK = 4: 20 KB of LDS and 64 VGPRs keep eight workgroups per CU (measured: the same code
// with 3 to 5 workgroups per CU is 8-40 % slower).  PMC: 1.48 G VALU instructions per launch
// (2.19 G for the per-record butterfly it replaces), the VALU pipes are busy for the whole kernel.
//
// Cost split of the backward at SYNTH-1M (ablations on a frozen scene, tools/experiments/abl.sh): phase 1 + 2 arithmetic
// 1.25 ms, staging + flush 0.9 ms, gather 0.27 ms.  The non-arithmetic part is a per-workgroup chain of dependent
// loads (id -> record, id -> slot base) behind barriers; it is neither shortened by more workgroups per CU, nor by
// slots addressed by sorted position plus an index indirection, nor by spreading staging and flush over all four
// waves with the fetches prefetched a round ahead (+5 %: the three extra waves issue the staging code too, and issue
// slots are what the kernel is short of).  What did help (-5.6 %): a 40-byte slot with the stamp inside (9 sums + stamp,
// five 8-byte stores) instead of 48 bytes plus a side array of stamps.  Compiled without packed-fp32 code generation.
//
// Arithmetic note.  sigma is evaluated as P = dx*(qa*dx + qb*dy) + qc*dy*dy with
// (qa,qb,qc) = -log2(e) * (a/2, b, c/2) folded at staging time, so exp(-sigma) = exp2(P) is
// one v_exp_f32.  Forward and backward use the identical expression, hence identical
// include/skip decisions.
#include <type_traits>

#include "common.h"
#include "tile_rect.h"
#include "blend_common.h"



struct TileGeom {
    int lb, cam, i, j, start, end, tx0, ty0;
    bool inside;
    float px, py;
};

__device__ __forceinline__ TileGeom tile_geom(int C, int W, int H, int tile_w, int tile_h,
                                              const int32_t* __restrict__ offsets, int n_isects) {
    TileGeom g;
    const int n_tiles = tile_w * tile_h, total = C * n_tiles;
        g.lb = xcd_remap(blockIdx.x, total, XCD_GROUP(C, n_tiles, tile_w));
    g.cam = g.lb / n_tiles;
    const int tile = g.lb - g.cam * n_tiles;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    g.tx0 = tx * 16; g.ty0 = ty * 16;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    g.j = g.tx0 + ((w & 1) << 3) + (lane & 7);
    g.i = g.ty0 + ((w >> 1) << 3) + (lane >> 3);
    g.inside = (g.i < H) && (g.j < W);
    g.px = (float)g.j + 0.5f; g.py = (float)g.i + 0.5f;
    g.start = offsets[g.lb];
    // n_isects < 0: the offsets array carries one more entry, the total (fused path: the count lives on the device)
    g.end = (g.lb == total - 1 && n_isects >= 0) ? n_isects : offsets[g.lb + 1];
    return g;
}

__device__ __forceinline__ int stage_record(const float4* __restrict__ splats, int64_t id, int t, int tx0, int ty0,
                                            float4* sR) {
    const float4 a = splats[id * 3 + 0];   // x y opacity conic.a
    const float4 b = splats[id * 3 + 1];   // conic.b conic.c r g
    const float4 c = splats[id * 3 + 2];   // b depth radius 0
    stage_qform(a, b, c, t, sR);
    // A quadrant (8x8 pixel centres) is relevant iff the ellipse {sigma(p) <= tau}, tau = ln(255 opacity), reaches
    // it: either the mean lies inside, or the minimum of sigma over one of its four edges is <= tau (sigma is
    // convex, so the minimum over the square sits on the boundary when the mean is outside).  tau is inflated
    // (2e-4 relative + 2e-4) so that every pixel of a quadrant declared irrelevant fails the alpha test of the
    // blend loop in float arithmetic as well.  Costs ~150 instructions per RECORD (one lane), and removes a
    // quarter of the per-(record, quadrant) wave iterations of the axis-aligned-box test it replaces.
    EllipseTest et;
    if (!ellipse_prepare(a.z, a.w, b.x, b.y, &et)) return 0;
    const float rx = ((float)tx0 + 0.5f) - a.x, ry = ((float)ty0 + 0.5f) - a.y;  // first pixel centre - mean
    int rel = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float dx0 = rx + (float)((q & 1) * 8), dy0 = ry + (float)((q >> 1) * 8);
        rel |= ellipse_hits_square(et, dx0, dx0 + 7.0f, dy0, dy0 + 7.0f) ? (1 << q) : 0;
    }
    return rel;
}

// TRAIN (the fused training calls): the per-trip bookkeeping of a saturating pixel -- threshold, alpha, T and index
// selects -- runs only in trips where some lane of the wave saturates (a wave-uniform branch that is almost never
// taken), and no "last contributing record" is tracked at all: last_ids receives (index of the record at which the
// pixel saturated) - 1, or INT_MAX for a pixel that never saturated.  The backward's test `record index <= last_ids`
// then admits exactly the records in front of the saturation point, and among those the alpha test alone decides, as
// it did in the forward.  Compare + select cost 1.7 ns each on this chip against 1.1 ns for a multiply or add
// (tools/probe/valu_cost.hip): the fast path drops six of them per (record, wave).
template <bool TRAIN>
__global__ __launch_bounds__(BLK) void k_blend_fwd(int C, int W, int H, int tile_w, int tile_h,
                                                   const float4* __restrict__ splats,
                                                   const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ flat, int n_isects,
                                                   float* __restrict__ out_rgb, float* __restrict__ out_alpha,
                                                   int32_t* __restrict__ last_ids,
                                                   uint64_t* __restrict__ cmask, int64_t cmask_words,
                                                   int32_t* __restrict__ tile_nb, int no_cull) {
    __shared__ float4 sR[BLK * 3];  // staged records (stage_qform)
    __shared__ uint64_t sMask[4][4];  // [quadrant][64-record chunk]
    const TileGeom g = tile_geom(C, W, H, tile_w, tile_h, offsets, n_isects);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Per-lane state is kept in VGPRs and updated with selects instead of branches: the scalar unit (one per
    // CU, shared by the 4 SIMDs) was the limiter of the branchy version (~38 SALU instructions of exec-mask
    // bookkeeping per record; PMC: SQ_INSTS_SALU ~ SQ_INSTS_VALU).
    // `thr` is the per-lane alpha threshold: 1/255 while the pixel is live, +inf once it saturated (or outside)
    float thr = g.inside ? 1.f / 255.f : __builtin_inff();
    float T = 1.0f, r = 0.f, gg = 0.f, b = 0.f;
    int cur = TRAIN ? 0x7fffffff : 0;
    int nb = 0;
    const int64_t mbase = mask_base(g.lb, g.start);
    for (int bs = g.start; bs < g.end; bs += BLK, ++nb) {
        if (__syncthreads_and(thr > 1.0f)) break;
        const int idx = bs + threadIdx.x;
        int rel = 0;
        if (idx < g.end) {
            rel = stage_record(splats, flat[idx], threadIdx.x, g.tx0, g.ty0, sR);
            if (no_cull) rel = 0xF;   // test hook: every staged record is walked by every wave
        }
        const uint64_t m0 = __ballot(rel & 1), m1 = __ballot(rel & 2), m2 = __ballot(rel & 4), m3 = __ballot(rel & 8);
        if (lane == 0) { sMask[0][w] = m0; sMask[1][w] = m1; sMask[2][w] = m2; sMask[3][w] = m3; }
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
            uint64_t m64 = uniform_u64(sMask[w][jj]);
            uint64_t contributed = 0;
            if (__builtin_amdgcn_ballot_w64(thr < 1.0f) == 0) m64 = 0;  // every pixel of this wave is saturated
            // the word is walked as two 32-bit halves: the scalar unit is shared by the CU's four SIMDs (one issue slot in
            // four cycles each) and 64-bit mask arithmetic costs it two instructions where 32-bit costs one
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
            uint32_t m = hh ? (uint32_t)(m64 >> 32) : (uint32_t)m64;
            uint32_t cont32 = 0;
            while (m) {
                const int bit = __builtin_ctz(m);
                m &= m - 1;
                const int t = jj * 64 + hh * 32 + bit;
                const float4 a = sR[3 * t];
                const float4 q = sR[3 * t + 1];
                const float cb = sR[3 * t + 2].x;
                const float dx = a.x - g.px, dy = a.y - g.py;
                const float P = blend_power(dx, dy, a.w, q.x, q.y);
                const float al0 = fminf(0.999f, a.z * __builtin_amdgcn_exp2f(P));
                // skipped: sigma < 0, or below the lane's threshold (visibility for a live pixel, +inf for a
                // saturated one) -- two independent compares and one select
                uint64_t okm = 0;
                bool ok = false;
                float al;
                if (TRAIN) {
                    okm = mask_not_positive(P) & mask_not_less(al0, thr);
                    al = zero_unless(okm, al0);
                } else {
                    ok = !(P > 0.f) && !(al0 < thr);
                    al = ok ? al0 : 0.f;
                }
                const float nT = T * (1.0f - al);        // == T exactly when al == 0
                const bool stop = nT <= 1e-4f;           // only a live pixel with al > 0 can get here (T > 1e-4 otherwise)
                float vis = al * T;
                uint64_t tm;
                if (TRAIN) {
                    // fast values first (same basic block as the compares: the ballot is their mask), then the fix-up
                    // of the rare trip in which some pixel of the wave saturates
                    tm = okm;                                    // ok <=> al > 0 (al0 >= thr > 0)
                    float Tn = nT;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(stop) != 0, 0)) {
                        thr = stop ? __builtin_inff() : thr;
                        vis = stop ? 0.0f : vis;                 // the record that saturates the pixel is not blended
                        Tn = stop ? T : nT;
                        cur = stop ? bs + t : cur;
                        tm = okm & ~__builtin_amdgcn_ballot_w64(stop);
                    }
                    T = Tn;
                } else {
                    thr = stop ? __builtin_inff() : thr;
                    al = stop ? 0.0f : al;                   // the record that saturates the pixel is not blended
                    vis = al * T;
                    T = stop ? T : nT;
                    const bool took = al > 0.0f;
                    cur = took ? bs + t : cur;
                    tm = __builtin_amdgcn_ballot_w64(took);
                }
                r = __builtin_fmaf(q.z, vis, r); gg = __builtin_fmaf(q.w, vis, gg); b = __builtin_fmaf(cb, vis, b);
                cont32 |= tm ? (1u << bit) : 0u;
            }
            contributed |= (uint64_t)cont32 << (32 * hh);
            }
            if (cmask && lane == 0) cmask[(int64_t)w * cmask_words + mbase + nb * 4 + jj] = contributed;
        }
    }
    if (g.inside) {
        const int64_t p = ((int64_t)g.cam * H + g.i) * W + g.j;
        out_rgb[3 * p] = r; out_rgb[3 * p + 1] = gg; out_rgb[3 * p + 2] = b;
        out_alpha[p] = 1.0f - T;
        last_ids[p] = TRAIN ? (cur == 0x7fffffff ? cur : cur - 1) : cur;
    }
    if (tile_nb && threadIdx.x == 0) tile_nb[g.lb] = nb;
}

// scratch for the forward->backward hand-off lives in the ctx
static int hand_off_buffers(st3r_ctx* ctx, int C, int tile_w, int tile_h, int64_t n_isects, uint64_t** cmask,
                            int64_t* words, int32_t** tile_nb) {
    const int64_t total = (int64_t)C * tile_w * tile_h;
    *words = (n_isects >> 6) + 4 * total + 8;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_CMASK, sizeof(uint64_t) * 4 * (size_t)*words, &p);
    if (rc) return rc;
    *cmask = (uint64_t*)p;
    rc = st3r_arena_get(ctx, SLOT_TILE_NB, sizeof(int32_t) * (size_t)total, &p);
    if (rc) return rc;
    *tile_nb = (int32_t*)p;
    return ST3R_OK;
}

int st3r_blend_fwd_cells_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                              const float* splats, const int32_t* offsets, const int32_t* flat, float* rgb, float* alpha,
                              int32_t* last_ids, uint64_t* cmask, int64_t cmask_words, int32_t* tile_nb);

int st3r_blend_fwd_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                        const float* splats, const int32_t* offsets, const int32_t* flat, int64_t n_isects,
                        float* rgb, float* alpha, int32_t* last_ids, bool for_backward, bool end_in_offsets) {
    const int total = C * tile_w * tile_h;
    uint64_t* cmask = nullptr; int64_t words = 0; int32_t* tile_nb = nullptr;
    if (for_backward) {
        int rc = hand_off_buffers(ctx, C, tile_w, tile_h, n_isects, &cmask, &words, &tile_nb);
        if (rc) return rc;
    }
    // the fused training calls (their own backward follows) take the TRAIN variant; debug flag 128 keeps them on the
    // gsplat-style bookkeeping (A/B)
    // the fused training calls blend with cell lists (gs_blend_cells.hip; debug flag 128 keeps them on the quadrant
    // kernel, flag 512 sends st3r_gs_render through the cell kernel as well: A/B and tests)
    if ((for_backward && end_in_offsets && !(ctx->debug_flags & 128)) || (end_in_offsets && (ctx->debug_flags & 512)))
        return st3r_blend_fwd_cells_impl(ctx, s, C, W, H, tile_w, tile_h, splats, offsets, flat, rgb, alpha, last_ids, cmask,
                                         words, tile_nb);
    if (for_backward && end_in_offsets)
        hipLaunchKernelGGL(k_blend_fwd<true>, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h, (const float4*)splats,
                           offsets, flat, -1, rgb, alpha, last_ids, cmask, words, tile_nb, ctx->debug_flags & 1);
    else
        hipLaunchKernelGGL(k_blend_fwd<false>, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h, (const float4*)splats,
                           offsets, flat, end_in_offsets ? -1 : (int)n_isects, rgb, alpha, last_ids, cmask, words, tile_nb,
                           ctx->debug_flags & 1);
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
    return st3r_blend_fwd_impl(ctx, (hipStream_t)stream, C, width, height, tile_w, tile_h, splats, offsets,
                               flatten_ids, n_isects, rgb, alpha, last_ids, true, false);
}

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
// Phase 2 of the backward pass: the wave's lanes turn from pixels into (record, pixel run) pairs.
//   lane = r + CHUNK * part: record r of the chunk, pixels part*CHUNK .. part*CHUNK + CHUNK-1 of the wave's 8x8
//   quadrant (a whole pixel row for CHUNK = 8, half a row for CHUNK = 4)
// Each lane walks its pixels, reading the pair scalars (g_o, fac) phase 1 left in LDS, and accumulates the pixel
// sums of its record in registers -- no cross-lane traffic per record.  The 64 / CHUNK partial sums of a record meet
// through the halving butterfly (once per CHUNK records) plus one or two DPP steps, and the result is stored (plain
// ds_write: a wave meets a record once per round) in the wave's own accumulator rows.
//   S_o = sum g_o, S_x = sum g_o dx, S_y = sum g_o dy, S_xx = sum g_o dx^2, S_xy, S_yy, S_rgb = sum fac v_rgb
// with g_o = vis dL/dalpha (so v_sigma = -opacity g_o; the record constants are applied at flush time).
__device__ __forceinline__ void bwd_phase2(const float2* __restrict__ pr, unsigned tpack, int cnt, int lane, const float4* sA, float* accw, float qxf_lane, float qyf_lane,
                                           const float (&pvr)[CHUNK], const float (&pvg)[CHUNK],
                                           const float (&pvb)[CHUNK]) {
    static_assert(CHUNK == 4, "one DPP row per record of the chunk");
    const int r = lane >> 4, part = lane & 15;
    wave_lds_sync();
    const int t = (tpack >> (8 * r)) & 0xFF;  // rows >= cnt read index 0 (valid); their sums are dropped below
    const float2 mean = *reinterpret_cast<const float2*>(&sA[t]);
    const float dy = mean.y - qyf_lane;
    const float2* src = pr + r * PAIR_STRIDE + PAIR_AT(part * CHUNK);
    // The lane's pixels are i = 0 .. CHUNK-1 to the right of its first one: with d0 = dx of the first pixel,
    // sum g dx = d0 W0 - W1 and sum g dx^2 = d0 (d0 W0 - 2 W1) + W2 for the index moments W_k = sum i^k g_i -- 12 VALU
    // instead of 21 for CHUNK = 4 (k_blend_bwd 2.27 -> 2.17 ms); the offsets are at most CHUNK-1 pixels, so the
    // cancellation costs a few ulps.
    float W0 = 0.f, W1 = 0.f, W2 = 0.f, Sr = 0.f, Sg = 0.f, Sb = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
        const float2 v = src[i];
        W0 += v.x;
        if (i == 1) { W1 = v.x; W2 = v.x; }
        if (i > 1) { W1 = fmaf((float)i, v.x, W1); W2 = fmaf((float)(i * i), v.x, W2); }
        Sr = fmaf(v.y, pvr[i], Sr); Sg = fmaf(v.y, pvg[i], Sg); Sb = fmaf(v.y, pvb[i], Sb);
    }
    const float d0 = mean.x - qxf_lane;
    const float So = W0, Sx = fmaf(d0, W0, -W1), Sxx = fmaf(d0, Sx - W1, W2);
    const float Sy = So * dy, Sxy = Sx * dy, Syy = Sy * dy;  // dy is the same for the lane's pixels
    float k0, k1, k2;
    reduce9_rows(Sx, Sy, So, Sxx, Sxy, Syy, Sr, Sg, Sb, k0, k1, k2);
    if (r < cnt && (lane & 3) == 0) {
        // bank b of the record's row -> slots: k0 -> {0,2,1,3}[b], k1 -> {4,6,5,7}[b], k2 -> 8 (bank 0)
        const int b = (lane >> 2) & 3;
        const int slot0 = ((b & 1) << 1) | (b >> 1);
        float* acc = accw + t * ACC_VALS;
        acc[slot0] = k0;
        acc[4 + slot0] = k1;
        if (b == 0) acc[8] = k2;
    }
    wave_lds_sync();
}

// TOUCH: `rects` is not a rectangle array but the per-pair touch stamps (uint32 [C*N]; only with `rectbase`, which leaves
// `rects` unused): a variant of its own, because the kernel sits at its 80-SGPR / 64-VGPR limits and SYNTH-1M never takes it
template <bool HAS_VA, bool TOUCH = false>  // v_alpha is NULL in the train step (the reference's loss ignores render_alpha, gs.py:126)
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(8))) void k_blend_bwd(int C, int W, int H, int tile_w, int tile_h,
                                                   const float4* __restrict__ splats,
                                                   const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ flat, int n_isects,
                                                   const float* __restrict__ out_alpha,
                                                   const int32_t* __restrict__ last_ids,
                                                   const float* __restrict__ v_rgb,
                                                   const float* __restrict__ v_alpha,
                                                   const uint64_t* __restrict__ cmask, int64_t cmask_words,
                                                   const int32_t* __restrict__ tile_nb,
                                                   const int32_t* __restrict__ cum,
                                                   const uint64_t* __restrict__ rects,
                                                   const uint64_t* __restrict__ rectbase, int tight,
                                                   float* __restrict__ vtile, int stamp, unsigned vt_cap) {

    // staged records, same q-form as the forward's but as three arrays (measured: the forward is faster with one
    // 48-byte record per staged index, this kernel with the split layout)
    __shared__ float4 sA[HB];   // x y opacity qa
    __shared__ float4 sB[HB];   // qb qc r g
    __shared__ float sC[HB];    // b
    __shared__ float sAccW[4][HB * ACC_VALS];             // per wave: the sums of the records it met this round
    __shared__ float2 sPair[4][CHUNK * PAIR_STRIDE];      // per wave: (g_o, fac) of CHUNK records x 64 pixels
    __shared__ uint64_t sClampW;                          // staged records that need the full tests (see `hard` below)
    const TileGeom g = tile_geom(C, W, H, tile_w, tile_h, offsets, n_isects);
    const int nb = tile_nb[g.lb];
    if (nb == 0) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p = ((int64_t)g.cam * H + g.i) * W + g.j;
    float T_final = 1.0f, vr = 0.f, vg = 0.f, vb = 0.f, va = 0.f;
    int bin_final = -1;
    if (g.inside) {
        T_final = 1.0f - out_alpha[p];
        vr = v_rgb[3 * p]; vg = v_rgb[3 * p + 1]; vb = v_rgb[3 * p + 2];
        if (HAS_VA) va = v_alpha[p];
        bin_final = last_ids[p];
    }
    // does any pixel of this wave bound the records it includes (saturated, outside the image, or gsplat-style
    // last_ids from the stand-alone forward)?
    const bool chk_index = __builtin_amdgcn_ballot_w64(bin_final != 0x7fffffff) != 0;
    // phase-2 view of the quadrant: this lane's CHUNK pixels are those of lanes pbase .. pbase + CHUNK-1; their
    // v_rgb go through the (still unused) chunk buffer into registers once per tile
    float2* pr = sPair[w];
    float* accw = sAccW[w];
    float pvr[CHUNK], pvg[CHUNK], pvb[CHUNK];
    const int pbase = (lane & 15) * CHUNK;      // (phase-2 lane = part + 16 * record: see bwd_phase2)
    {
        float* px = reinterpret_cast<float*>(pr);
        px[lane] = vr; px[64 + lane] = vg; px[128 + lane] = vb;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < CHUNK; ++i) { pvr[i] = px[pbase + i]; pvg[i] = px[64 + pbase + i]; pvb[i] = px[128 + pbase + i]; }
        wave_lds_sync();
    }
    const float qxf = (float)(g.tx0 + ((w & 1) << 3) + (pbase & 7)) + 0.5f;
    const float qyf_part = (float)(g.ty0 + ((w >> 1) << 3) + (pbase >> 3)) + 0.5f;
    // constants of the easy rounds' alpha test, kept in vector registers (a VOP3 instruction takes no literal, and a scalar
    // operand would cost the slot the form is there to save): 2^64 and -t' 2^64, t' = the float below 1/255
    float k_big = 0x1p64f, k_neg_thr = -__int_as_float(0x3b808080) * 0x1p64f;   // (1.f / 255.f is 0x3b808081)
    asm volatile("" : "+v"(k_big), "+v"(k_neg_thr));
    float T = T_final;
    // gsplat keeps buffer[k] = sum of the colours blended behind the current record; only its dot product
    // with the pixel's v_rgb is ever used, so one scalar replaces the three components
    float bv = 0.f;
    const int64_t mbase = mask_base(g.lb, g.start);
    // wave-uniform pointer (scalar loads); the mask word of a round is fetched one round ahead so that no dependent
    // load sits at the head of a round
    const uint64_t* wmask = cmask + (int64_t)__builtin_amdgcn_readfirstlane(w) * cmask_words + mbase;
    uint64_t m_next = wmask[(BLK / HB) * nb - 1];
    for (int hb = (BLK / HB) * nb - 1; hb >= 0; --hb) {
        const int bs = g.start + hb * HB;
        const int bsz = min(HB, g.end - bs);
        const uint64_t m_cur = m_next;
        if (hb > 0) m_next = wmask[hb - 1];
        if (bsz <= 0) continue;   // the tail of the last forward batch may be empty (uniform over the workgroup)
        // (no barrier here -- round 6: the staging arrays are free once every wave is past the walk of the previous round,
        // which the barrier in front of the flush established, and the staging wave is the one that flushed; the waves' own
        // accumulator rows are next written after the barrier below.  Two workgroup barriers per round instead of three.)
        // ---- staging: one record per thread (threads 0..HB-1).  A record some wave contributed to also fixes its
        // output slot now, so that the flush below is loads-free:  u = cum_excl[pid] + index of this tile inside the
        // record's tile rectangle (same float ops as the emit kernel => same integers)
        int my_u = -1, my_cb = 0;
        float my_op = 0.f, my_ca = 0.f, my_cbb = 0.f, my_cc = 0.f;
        if ((int)threadIdx.x < bsz) {
            const int t = threadIdx.x;
            const int64_t my_id = flat[bs + t];
            const int64_t word = mbase + hb;   // HB = 64: one (wave-uniform) mask word per round and wave
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) my_cb |= (int)((cmask[ww * cmask_words + word] >> (t & 63)) & 1ull) << ww;
            // only a record some wave contributed to is ever read from the staging arrays: the others' 48 bytes are not
            // fetched (their slots of sA / sB / sC keep whatever an earlier round left there)
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
            if (my_cb)
            {
                a = splats[my_id * 3 + 0];   // x y opacity conic.a
                b = splats[my_id * 3 + 1];   // conic.b conic.c r g
                c = splats[my_id * 3 + 2];   // b depth radius 0
                sA[t] = make_float4(a.x, a.y, a.z, -0.5f * LOG2E * a.w);
                sB[t] = make_float4(-LOG2E * b.x, -0.5f * LOG2E * b.y, b.z, b.w);
                sC[t] = c.x;
            }
            // Rounds whose records are all "easy" walk without two of the per-pixel tests (wave-uniform choice below):
            //   * opacity * exp(-sigma) can only exceed the 0.999 clamp when the opacity does (sigma >= 0 for every
            //     included pixel);
            //   * the test P > 0 (gsplat's sigma < 0) can only fire for a conic that is not safely positive definite: with
            //     rho = |b| / sqrt(a c), -Q >= (1 - rho) (|qa| dx^2 + |qc| dy^2) for the exact form Q, while the five
            //     roundings of blend_power move P by less than 4 * 2^-24 of that sum; 1 - rho^2 >= 2e-3 leaves a factor
            //     of several thousand.  (Such a conic has an axis ratio above ~30: needles.)
            // Both are properties of the RECORD, decided here once instead of per pixel and trip.
            const float det_ = a.w * b.y - b.x * b.x;
            // (0.998, not 0.999: v_exp_f32 is good to an ulp, and opacity * exp2(P) must stay below the clamp whatever it returns)
            const bool hard = my_cb && !(a.z <= 0.998f && a.w > 0.f && b.y > 0.f && det_ >= 2e-3f * (a.w * b.y));
            const uint64_t cw = __builtin_amdgcn_ballot_w64(hard);
            if (t == 0) sClampW = cw;
            if (my_cb && rectbase) {   // fused path: slot base and rectangle in one gathered word
                // (scenes of many slots per pair: the pair is marked as touched by this backward call -- the gather skips
                // the slot ranges of pairs nobody marked; a benign race, every writer stores the same stamp)
                if (TOUCH) reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(rects))[my_id] = (uint32_t)stamp;
                const uint64_t r = rectbase[my_id];
                const int x0 = (int)(r & 0x3FF), y0 = (int)((r >> 10) & 0x3FF), rw = (int)((r >> 20) & 0x3FF);
                my_u = (int)(r >> 32) + ((g.ty0 >> 4) - y0) * rw + ((g.tx0 >> 4) - x0);
                my_op = a.z; my_ca = a.w; my_cbb = b.x; my_cc = b.y;
            } else if (my_cb) {
                int x0, y0, rw;
                if (rects) {   // fused path: the packed rectangle the emit kernel used (one 8-byte load instead of
                               // ~100 instructions of tight_tile_rect)
                    const uint64_t r = rects[my_id];
                    x0 = (int)(r & 0xFFFF); y0 = (int)((r >> 16) & 0xFFFF); rw = (int)((r >> 32) & 0xFFFF);
                } else {
                    TileRect tr = ref_tile_rect(a.x, a.y, (float)__float_as_int(c.z), 16, tile_w, tile_h);
                    if (tight) tr = tight_tile_rect(tr, a.x, a.y, a.z, a.w, b.x, b.y);
                    x0 = tr.x0; y0 = tr.y0; rw = tr.x1 - tr.x0;
                }
                const int cum_excl = my_id == 0 ? 0 : cum[my_id - 1];
                my_u = cum_excl + ((g.ty0 >> 4) - y0) * rw + ((g.tx0 >> 4) - x0);
                my_op = a.z; my_ca = a.w; my_cbb = b.x; my_cc = b.y;
            }
        }
        __syncthreads();
        // ---- phase 1: lanes are pixels; up to CHUNK records, back to front (unrolled: the chunk row is an immediate
        // offset, the records' staged indices travel to phase 2 in one scalar, 8 bits each).  Compares and selects cost
        // 1.7 ns each on this chip against 1.1 ns for an add or multiply (tools/probe/valu_cost.hip), so the two tests
        // that almost never matter are compiled out of the rounds that do not need them (wave-uniform choice):
        //   CHK    the record-index bound: only waves holding a saturated (or outside) pixel need it -- every other
        //          pixel's last_ids is INT_MAX (fused path) and the alpha test alone decides, as in the forward;
        //   CLAMP  the 0.999 clamp: only rounds that stage a record with opacity > 0.999 (sClampW).
        // What goes to phase 2 is g' = opacity * g_o (alpha * dL/dalpha instead of vis * dL/dalpha): the flush divides
        // the one sum that needs it.
        auto walk = [&](auto chk_tag, auto clamp_tag) {
            constexpr bool CHK = decltype(chk_tag)::value, CLAMP = decltype(clamp_tag)::value;
            uint64_t m = m_cur;   // HB = 64: one mask word per round
            while (m) {
                unsigned tpack = 0;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < CHUNK; ++k) {
                    if (m) {
                        const int t = 63 - __builtin_clzll(m);
                        m &= ~(1ull << t);
                        const float4 a = sA[t];
                        const float4 q = sB[t];
                        const float cb_ = sC[t];
                        const float dx = a.x - g.px, dy = a.y - g.py;
                        const float P = blend_power(dx, dy, a.w, q.x, q.y);
                        const float vis0 = __builtin_amdgcn_exp2f(P);
                        const float ov0 = a.z * vis0;
                        float alpha;
                        if (CLAMP) {
                            const float al0 = fminf(0.999f, ov0);
                            // same include test as the forward pass.  Branch-free: a lane that does not include this record
                            // gets alpha = 0, hence 1/(1-alpha) = 1, fac = 0, g' = 0.
                            uint64_t okm = mask_not_positive(P) & mask_not_less(al0, 1.f / 255.f);
                            if (CHK) okm &= __builtin_amdgcn_ballot_w64(bs + t <= bin_final);
                            alpha = zero_unless(okm, al0);
                        } else {
                            // easy round: P <= 0 and alpha <= 0.999 hold by construction; what remains is alpha >= 1/255 --
                            // as arithmetic (a compare + select pair is two four-cycle instructions and two scalar-port
                            // slots, a multiply-add + multiply are two two-cycle ones): step = clamp01((alpha - t') 2^64) with
                            // t' the float below 1/255 is exactly 1 for alpha >= 1/255 and exactly 0 below
                            float step;
                            asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(step) : "v"(ov0), "v"(k_big), "v"(k_neg_thr));
                            alpha = ov0 * step;
                            if (CHK) alpha = zero_unless(__builtin_amdgcn_ballot_w64(bs + t <= bin_final), alpha);
                        }
                        // a clamped alpha (opacity*vis > 0.999) passes no gradient to sigma / opacity
                        float alpha_u = alpha;
                        if (CLAMP) alpha_u = ov0 <= 0.999f ? alpha : 0.f;
                        const float ra = __builtin_amdgcn_rcpf(1.0f - alpha);
                        const float cv = q.z * vr + q.w * vg + cb_ * vb;   // colour . v_rgb
                        T *= ra;
                        const float fac = alpha * T;
                        float v_al = cv * T - bv * ra;
                        if (HAS_VA) v_al += T_final * ra * va;
                        bv += cv * fac;
                        pr[k * PAIR_STRIDE + PAIR_AT(lane)] = make_float2(alpha_u * v_al, fac);
                        tpack |= (unsigned)t << (8 * k);
                        cnt = k + 1;
                    }
                }
                bwd_phase2(pr, tpack, cnt, lane, sA, accw, qxf, qyf_part, pvr, pvg, pvb);
            }
        };
        const bool clamp_round = (uniform_u64(sClampW) & m_cur) != 0;
        if (chk_index) {
            if (clamp_round) walk(std::true_type{}, std::true_type{}); else walk(std::true_type{}, std::false_type{});
        } else {
            if (clamp_round) walk(std::false_type{}, std::true_type{}); else walk(std::false_type{}, std::false_type{});
        }
        __syncthreads();
        // ---- flush: the (at most four) wave sums of a record -> its stamped slot in HBM.  Slot indices come from the
        // scan over the TRUE tile counts; in an asynchronous step that outgrew its capacity they can exceed the slots the
        // buffer has (that step is discarded anyway): such a record is not written (unsigned: a wrapped index too)
        if (my_cb && (unsigned)my_u < vt_cap) {
            float acc[ACC_VALS];
#pragma unroll
            for (int k = 0; k < ACC_VALS; ++k) acc[k] = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                if (my_cb & (1 << ww)) {
#pragma unroll
                    for (int k = 0; k < ACC_VALS; ++k) acc[k] += sAccW[ww][threadIdx.x * ACC_VALS + k];
                }
            }
            // record constants: v_sigma = -opacity g_o;  v_mean2d = v_sigma (a dx + b dy, b dx + c dy);
            // v_conic = v_sigma (dx^2 / 2, dx dy, dy^2 / 2)
            // (the six geometric sums arrive multiplied by the opacity: phase 1 hands over alpha * dL/dalpha)
            const float sx = -acc[0], sy = -acc[1];
            float2* dst = reinterpret_cast<float2*>(vtile + (int64_t)my_u * VT_STRIDE);
            dst[0] = make_float2(my_ca * sx + my_cbb * sy, my_cbb * sx + my_cc * sy);
            dst[1] = make_float2(my_op != 0.f ? acc[2] / my_op : 0.f, -0.5f * acc[3]);
            dst[2] = make_float2(-acc[4], -0.5f * acc[5]);
            dst[3] = make_float2(acc[6], acc[7]);
            dst[4] = make_float2(acc[8], __int_as_float(stamp));
        }
    }
}

// v_splats[pid] = sum over the pair's tiles of the slots stamped by this backward call, in slot order (deterministic
// given the slots).  A workgroup owns 256 consecutive (camera, gaussian) pairs, whose slots are one contiguous range.
// The range is streamed through LDS 256 slots at a time with lane = slot (coalesced reads, no per-lane trip counts on
// the HBM side; the next 256 slots are in flight while the current ones are summed); then lane = pair adds the rows
// of its own slots in slot order.  Measured at SYNTH-1M: 0.26 ms, one thread per pair walking its slots 0.35 ms.
__global__ __launch_bounds__(256) void k_gather_vtile(int64_t n_pairs, const int32_t* __restrict__ cum,
                                                      const float* __restrict__ vtile, int stamp, unsigned vt_cap,
                                                      float4* __restrict__ v_splats) {
    constexpr int ROW = ACC_VALS;   // odd stride: rows of neighbouring slots fall into different banks
    __shared__ int sCum[257];
    __shared__ float sVal[256 * ROW];
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * 256;
    const int np = (int)min((int64_t)256, n_pairs - p0);
    if (tid == 0) sCum[0] = p0 == 0 ? 0 : cum[p0 - 1];
    sCum[tid + 1] = cum[p0 + min(tid, np - 1)];
    __syncthreads();
    const int s0 = sCum[0], s1 = sCum[np];
    const int my_start = sCum[tid], my_end = tid < np ? sCum[tid + 1] : sCum[tid];
    float acc[ACC_VALS];
#pragma unroll
    for (int k = 0; k < ACC_VALS; ++k) acc[k] = 0.f;
    float2 q0, q1, q2, q3, q4;
    auto fetch = [&](int u) {
        q0 = q1 = q2 = q3 = q4 = make_float2(0.f, 0.f);   // stamp 0 = never written
        if (u < s1 && (unsigned)u < vt_cap) {   // (vt_cap: see the flush of k_blend_bwd)
            const float2* src = reinterpret_cast<const float2*>(vtile + (int64_t)u * VT_STRIDE);
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; q4 = src[4];
        }
    };
    fetch(s0 + tid);
    for (int base = s0; base < s1; base += 256) {
        const bool live = __float_as_int(q4.y) == stamp;
        float* row = sVal + tid * ROW;
        row[0] = live ? q0.x : 0.f; row[1] = live ? q0.y : 0.f; row[2] = live ? q1.x : 0.f;
        row[3] = live ? q1.y : 0.f; row[4] = live ? q2.x : 0.f; row[5] = live ? q2.y : 0.f;
        row[6] = live ? q3.x : 0.f; row[7] = live ? q3.y : 0.f; row[8] = live ? q4.x : 0.f;
        __syncthreads();
        fetch(base + 256 + tid);
        const int lo = max(my_start, base) - base, hi = min(my_end, base + 256) - base;
        for (int r = lo; r < hi; ++r) {
            const float* src = sVal + r * ROW;
#pragma unroll
            for (int k = 0; k < ACC_VALS; ++k) acc[k] += src[k];
        }
        __syncthreads();
    }
    if (tid < np) {
        const int64_t pid = p0 + tid;
        v_splats[pid * 3 + 0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        v_splats[pid * 3 + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        v_splats[pid * 3 + 2] = make_float4(acc[8], 0.f, 0.f, 0.f);
    }
}

int st3r_blend_bwd_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                        const float* splats, const int32_t* offsets, const int32_t* flat, int64_t n_isects,
                        const float* alpha, const int32_t* last_ids, const float* v_rgb, const float* v_alpha,
                        const int32_t* cum, const uint64_t* rects, const uint64_t* rectbase, int tight, int64_t n_pairs,
                        float* v_splats, bool end_in_offsets, st3r_vtile_ref* defer) {
    // defer != NULL: the caller's next kernel sums the slots per pair itself (gs_project_bwd.hip); v_splats is not written
    if (defer) *defer = st3r_vtile_ref{cum, nullptr, 0, 0u, nullptr};
    if (n_isects == 0) {
        if (!defer) HIP_TRY(hipMemsetAsync(v_splats, 0, sizeof(float) * ST3R_SPLAT_STRIDE * (size_t)n_pairs, s));
        return ST3R_OK;   // (defer: every pair's slot range is empty -- `cum` is all zeros -- and vt_cap = 0 guards the rest)
    }
    uint64_t* cmask; int64_t words; int32_t* tile_nb;
    int rc = hand_off_buffers(ctx, C, tile_w, tile_h, n_isects, &cmask, &words, &tile_nb);
    if (rc) return rc;
    // per-(record, tile) partial gradients: 9 floats + a stamp; a slot counts only if its stamp equals this
    // call's, so the buffer is never cleared per call (the stamps are zeroed when the buffer is (re)allocated and
    // when the counter is about to wrap)
    void* p; int grown = 0;
    rc = st3r_arena_get2(ctx, SLOT_VTILE, sizeof(float) * VT_STRIDE * (size_t)n_isects, &p, &grown);
    if (rc) return rc;
    float* vtile = (float*)p;
    // Per-pair "touched" stamps, only where they pay: a scene with many slots per pair (large Gaussians; more than eight
    // per pair SLOT of the call -- the gather's item-parallel form, which is the one that skips, starts at six per visible pair
    // of a wave, and SYNTH-1M late in training reaches five) AND a gather that follows in the projection backward.  There most slots belong to pairs that never contribute
    // (behind the saturation depth of their tiles): the configs[1] example reads 6.4 GB of slots per step of which ~15 % were
    // written.  SYNTH-1M (3.3 slots per pair) does not take this path: one more scattered store per staged record for nothing.
    uint32_t* touch = nullptr;
    int grown_t = 0;
    if (defer && rectbase && !v_alpha && n_isects > 8 * n_pairs) {
        void* pt;
        rc = st3r_arena_get2(ctx, SLOT_PAIR_TOUCH, sizeof(uint32_t) * (size_t)n_pairs, &pt, &grown_t);
        if (rc) return rc;
        touch = (uint32_t*)pt;
    }
    if (grown || ctx->bwd_stamp >= 2147483000) {   // (the stamps restart: both stamp stores are cleared together)
        HIP_TRY(hipMemsetAsync(p, 0, ctx->slot_bytes[SLOT_VTILE], s));
        if (ctx->slot_ptr[SLOT_PAIR_TOUCH])
            HIP_TRY(hipMemsetAsync(ctx->slot_ptr[SLOT_PAIR_TOUCH], 0, ctx->slot_bytes[SLOT_PAIR_TOUCH], s));
        ctx->bwd_stamp = 0;
    } else if (grown_t) {   // (a fresh touch array: zeros are older than every stamp in use)
        HIP_TRY(hipMemsetAsync(touch, 0, ctx->slot_bytes[SLOT_PAIR_TOUCH], s));
    }
    const int stamp = ++ctx->bwd_stamp;
    const int total = C * tile_w * tile_h;
    const unsigned vt_cap = (unsigned)(ctx->slot_bytes[SLOT_VTILE] / (sizeof(float) * VT_STRIDE));
    if (v_alpha)
        hipLaunchKernelGGL(k_blend_bwd<true>, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h,
                           (const float4*)splats, offsets, flat, end_in_offsets ? -1 : (int)n_isects, alpha, last_ids, v_rgb,
                           v_alpha, cmask, words, tile_nb, cum, rects, rectbase, tight, vtile, stamp, vt_cap);
    else if (touch)
        hipLaunchKernelGGL((k_blend_bwd<false, true>), dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h,
                           (const float4*)splats, offsets, flat, end_in_offsets ? -1 : (int)n_isects, alpha, last_ids, v_rgb,
                           v_alpha, cmask, words, tile_nb, cum, reinterpret_cast<const uint64_t*>(touch), rectbase, tight, vtile,
                           stamp, vt_cap);
    else
        hipLaunchKernelGGL(k_blend_bwd<false>, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h,
                           (const float4*)splats, offsets, flat, end_in_offsets ? -1 : (int)n_isects, alpha, last_ids, v_rgb,
                           v_alpha, cmask, words, tile_nb, cum, rects, rectbase, tight, vtile, stamp, vt_cap);
    LAUNCH_CHECK();
    if (defer) { *defer = st3r_vtile_ref{cum, vtile, stamp, vt_cap, touch}; return ST3R_OK; }
    hipLaunchKernelGGL(k_gather_vtile, dim3(ceil_div(n_pairs, 256)), dim3(256), 0, s, n_pairs, cum, vtile, stamp, vt_cap,
                       (float4*)v_splats);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_blend_bwd(st3r_ctx* ctx, void* stream, int C, int width, int height, int tile_size,
                                  int tile_w, int tile_h, const float* splats, const int32_t* offsets,
                                  const int32_t* flatten_ids, int64_t n_isects, const float* alpha,
                                  const int32_t* last_ids, const float* v_rgb, const float* v_alpha,
                                  const int32_t* cum_tiles, int64_t n_pairs, float* v_splats) {
    ARG_CHECK(ctx && C > 0 && width > 0 && height > 0 && tile_size == 16);
    ARG_CHECK(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16);
    ARG_CHECK(splats && offsets && alpha && last_ids && v_rgb && v_splats && cum_tiles && n_pairs >= 0);
    ARG_CHECK(n_isects >= 0 && n_isects < 2147483647LL && (n_isects == 0 || flatten_ids));
    return st3r_blend_bwd_impl(ctx, (hipStream_t)stream, C, width, height, tile_w, tile_h, splats, offsets,
                               flatten_ids, n_isects, alpha, last_ids, v_rgb, v_alpha, cum_tiles, nullptr, nullptr, 0, n_pairs,
                               v_splats, false, nullptr);
}
