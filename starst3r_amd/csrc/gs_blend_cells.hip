// Per-tile alpha blending with 4x4-pixel CELL lists -- the formulation of the fused training path (round 3).
// Replaces gsplat rasterize_to_pixels fwd/bwd (starster/gs.py:76 and the autograd pass of starster/gs.py:153), like
// gs_blend.hip, whose quadrant formulation stays behind the stand-alone entry points.
//
// Why.  In the quadrant formulation a wave owns an 8x8 quadrant and meets one record per trip; a record's
// alpha >= 1/255 footprint covers ~30 % of the 64 pixels, and every trip pays ~6 ns of scalar bit scanning on a scalar
// unit that four SIMDs share (tools/probe/trip_replay.hip: 43.4 ns per trip, 37.2 without the scan, 28.7 without scan
// and LDS reads).  Here a wave still owns an 8x8 quadrant, but each of its four 16-lane rows owns one 4x4 CELL and
// walks its OWN list of records: a trip blends up to four different records, one per row.  The lists are built once
// per batch of 255 records by the staging threads:
//   cell test   the record's ellipse {alpha >= 1/255} against the sixteen cells, exactly, by row strips: for the four
//               pixel-centre rows of a cell row the ellipse's x extent has a closed form (a concave / convex function of
//               dy evaluated at its clamped extremum), which gives the hit cells of that row as a bit range -- ~40
//               instructions per strip, no loops over cells (tile_rect.h: cell_mask16);
//   positions   the sixteen hit bits become sixteen byte counters in four registers, one DPP prefix scan of the four
//               words over the wave gives every record its position in each cell's list (255 records: a byte holds a
//               whole list length), the wave totals meet in LDS;
//   lists       every hit (record, cell) stores the record's LDS byte offset as a 16-bit list entry; lists are padded
//               with the offset of an all-zero sentinel record up to the longest list of the wave that walks them.
// A trip then costs one ds_read_u16 (row-uniform address) more than before and no scalar work beyond the loop
// counter; the trip count of a wave is the longest of its four lists (tools/cell_stats.py: 0.745 of the quadrant
// trips at SYNTH-1M, 0.65 if the four lists were always equally long), re-evaluated every 32 list positions over the
// rows that still hold a live pixel (0.684).
// The alpha / saturation tests are exec-masked instead of select-based (lanes that fail skip the updates; a trip
// that no lane passes skips the body: -10 % in the replay probe).
//
// Forward -> backward hand-off: exactly the quadrant forward's (gs_blend.hip) -- per pixel T_final and the saturation
// index, per tile the number of batches of 256 records, and per (wave, 64-record chunk) the word of records that
// contributed to some pixel of the wave's quadrant -- so the backward kernel is shared.  The contribution bits are
// collected without scalar work: a lane that blends the record of list position k sets bit k of a register (one v_or
// inside the exec-masked body); every 32 trips the row ORs its lanes' words (four DPP steps) and lane idx stores a flag
// byte (one per wave and record) for the positions 2 idx, 2 idx + 1 if their bits are set; when the batch is done the
// staging thread of a record turns its four flag bytes into four ballots.  (First attempt: every blending lane stored
// the flag byte itself -- sixteen lanes, one address: the LDS serialises such stores, +0.15 ms.)
#include "blend_common.h"
#include <type_traits>

#define CB 256                  // records per batch (= the quadrant forward's: the contribution words line up)
#define CL_CAP 264              // list capacity per cell in entries (256 + the read-ahead of the trip loop)
#define SLOT 48                 // LDS bytes per staged record: x y opacity qa | qb qc r g | b - - -
#define SENT_OFF (CB * SLOT)    // LDS byte offset of the sentinel record (opacity 0: fails every alpha test)

// ---- per-wave list building.  The staging threads leave the 16-bit cell masks of the batch's records in LDS (sCm[t]);
// after the barrier every WAVE builds the four lists of its own cells, so no further workgroup barrier is needed:
//   lane (row rho = cell, idx)   owns the 16 records [16 idx, 16 idx + 16): two 16-byte reads, the cell's bit of each
//                                mask gathered into a 16-bit piece (v_bfe + v_lshl_or per record),
//   positions                    popcount, then a prefix scan over the 16 lanes of the row (four DPP steps),
//   entries                      the row first fills its whole list with the sentinel offset (two 16-byte stores per
//                                lane), then every lane writes the LDS byte offsets of its hit records at its positions
//                                (a loop over the set bits of the piece: as many trips as the fullest piece of the wave).
// Returns the lengths of the wave's four lists packed in one word (a byte each would not hold 256: 9 bits x 4 do not fit
// either, so two words): lens.x = l0 | l1 << 16, lens.y = l2 | l3 << 16.
// (Round 3, first version: one thread per record scattered into sixteen lists after a packed-counter prefix scan over
// the workgroup -- ~280 instructions per wave and batch and a second barrier; this form costs ~110.)
__device__ __forceinline__ uint2 build_wave_lists(const uint16_t* sCm, uint16_t* myList, int cell, int lane) {
    const int idx = lane & 15;
    const uint4 d0 = *reinterpret_cast<const uint4*>(sCm + 16 * idx);
    const uint4 d1 = *reinterpret_cast<const uint4*>(sCm + 16 * idx + 8);
    // prefill: 256 entries = 512 bytes per list, 32 bytes per lane
    const unsigned sent2 = (unsigned)SENT_OFF | ((unsigned)SENT_OFF << 16);
    const uint4 sv = make_uint4(sent2, sent2, sent2, sent2);
    *reinterpret_cast<uint4*>(myList + 16 * idx) = sv;
    *reinterpret_cast<uint4*>(myList + 16 * idx + 8) = sv;
    const unsigned dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    unsigned piece = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        piece |= ((dw[j] >> cell) & 1u) << (2 * j);
        piece |= ((dw[j] >> (cell + 16)) & 1u) << (2 * j + 1);
    }
    const unsigned cnt = __builtin_popcount(piece);
    unsigned inc = cnt;
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                 : "+v"(inc));
    const int l0 = __builtin_amdgcn_readlane((int)inc, 15), l1 = __builtin_amdgcn_readlane((int)inc, 31);
    const int l2 = __builtin_amdgcn_readlane((int)inc, 47), l3 = __builtin_amdgcn_readlane((int)inc, 63);
    wave_lds_sync();   // the prefill is ordered before the entries (same wave: LDS operations complete in order)
    uint16_t* dst = myList + (inc - cnt);
    unsigned off = (unsigned)idx * (16u * SLOT);
    while (piece) {
        const int bpos = __builtin_ctz(piece);
        piece &= piece - 1;
        *dst++ = (uint16_t)(off + (unsigned)bpos * SLOT);
    }
    wave_lds_sync();
    return make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
}

struct CellPx {
    float px, py, thr, T, r, g, b;
    int cur;
};

// one trip of one lane: the record (a, q, cb) staged at LDS byte offset `off`.  Same arithmetic, operation for
// operation, as the quadrant kernel (blend_power / explicit fma: the images are bit-identical, tests), but exec-masked:
// lanes that fail the alpha test skip the updates, a trip that no lane passes skips the body.  The saturation
// bookkeeping sits behind a wave-uniform branch that is almost never taken.
// LEAN (round 5): the batch holds only records whose conic is safely positive definite and whose opacity is <= 0.998 (decided
// per record while staging, see `hard` in the kernel): P <= 0 holds for every pixel and opacity * exp2(P) cannot reach the clamp,
// so the trip drops gsplat's `sigma < 0` test and the min -- two four-cycle instructions of ~25 (tools/probe/valu_issue.hip);
// the values are bit-identical to the full trip's.
template <bool LEAN>
__device__ __forceinline__ void cell_trip(const float4 a, const float4 q, const float cb, int off, int bs, CellPx& s,
                                          unsigned& cbits, unsigned bit) {
    const float dx = a.x - s.px, dy = a.y - s.py;
    const float P = blend_power(dx, dy, a.w, q.x, q.y);
    const float ov = a.z * __builtin_amdgcn_exp2f(P);
    const float al = LEAN ? ov : fminf(0.999f, ov);
    if ((LEAN || !(P > 0.f)) && !(al < s.thr)) {
        const float nT = s.T * (1.0f - al);
        float vis = al * s.T, Tn = nT;
        const bool stop = nT <= 1e-4f;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(stop) != 0, 0)) {
            // the record that would saturate a pixel is not blended; the pixel is done (in-place fix-up of the fast values)
            s.thr = stop ? __builtin_inff() : s.thr;
            vis = stop ? 0.0f : vis;
            Tn = stop ? s.T : nT;
            s.cur = stop ? bs + (int)(((unsigned)off >> 4) * 43691u >> 17) : s.cur;   // off / 48
        }
        // this lane took part in the record of this list position (a lane that only saturates on it counts too: a superset
        // costs the backward a trip whose sums are zero, never a wrong sum)
        cbits |= bit;
        s.T = Tn;
        s.r = __builtin_fmaf(q.z, vis, s.r); s.g = __builtin_fmaf(q.w, vis, s.g); s.b = __builtin_fmaf(cb, vis, s.b);
    }
}

struct CellTile {
    int lb, cam, tx0, ty0, start, end, i, j, cell;
    bool inside;
};

// lane -> pixel: wave w owns quadrant (w & 1, w >> 1), its 16-lane row rho the cell (rho & 1, rho >> 1) of the
// quadrant, lane idx of the row the pixel (idx & 3, idx >> 2) of the cell
__device__ __forceinline__ CellTile cell_tile(int C, int W, int H, int tile_w, int tile_h, const int32_t* __restrict__ offsets) {
    CellTile g;
    const int n_tiles = tile_w * tile_h, total = C * n_tiles;
    g.lb = xcd_remap(blockIdx.x, total, XCD_GROUP(C, n_tiles, tile_w));
    g.cam = g.lb / n_tiles;
    const int tile = g.lb - g.cam * n_tiles;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    g.tx0 = tx * 16; g.ty0 = ty * 16;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, rho = lane >> 4, idx = lane & 15;
    const int cx = ((w & 1) << 1) | (rho & 1), cy = ((w >> 1) << 1) | (rho >> 1);
    g.cell = cy * 4 + cx;
    g.j = g.tx0 + 4 * cx + (idx & 3);
    g.i = g.ty0 + 4 * cy + (idx >> 2);
    g.inside = (g.i < H) && (g.j < W);
    g.start = offsets[g.lb];
    g.end = offsets[g.lb + 1];   // the fused path's table carries the total as its last entry
    return g;
}

__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(7))) void k_blend_fwd_cells(int C, int W, int H, int tile_w, int tile_h,
                                                         const float4* __restrict__ splats,
                                                         const int32_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ flat,
                                                         float* __restrict__ out_rgb, float* __restrict__ out_alpha,
                                                         int32_t* __restrict__ last_ids,
                                                         uint64_t* __restrict__ cmask, int64_t cmask_words,
                                                         int32_t* __restrict__ tile_nb, int no_cull) {
    __shared__ float4 sR[(CB + 1) * (SLOT / 16)];   // staged records (q-form) and the sentinel
    __shared__ uint16_t sList[16][CL_CAP];          // per cell: LDS byte offsets of its records, front to back
    __shared__ uint16_t sCm[BLK];                   // the staged records' cell masks
    __shared__ unsigned sFlag[CB];                  // per staged record: byte w != 0 <=> wave w contributed to it
    __shared__ int sHard[2];                        // batch nb holds a record that needs the full trip (slot nb & 1)
    const CellTile g = cell_tile(C, W, H, tile_w, tile_h, offsets);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    CellPx s;
    s.px = (float)g.j + 0.5f; s.py = (float)g.i + 0.5f;
    s.thr = g.inside ? 1.f / 255.f : __builtin_inff();
    s.T = 1.0f; s.r = s.g = s.b = 0.f; s.cur = 0x7fffffff;
    if (threadIdx.x < SLOT / 16) sR[CB * (SLOT / 16) + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 2) sHard[threadIdx.x] = 0;
    char* sRb = reinterpret_cast<char*>(sR);
    uint16_t* lp = sList[g.cell];
    unsigned char* my_flags = reinterpret_cast<unsigned char*>(sFlag) + w;
    const int64_t mbase = mask_base(g.lb, g.start);
    // the flag bytes of the finished batch nbp -> four contribution words per staging wave (64 records each)
    auto flags_out = [&](int nbp) {
        const unsigned f = sFlag[threadIdx.x];
        const uint64_t c0 = __builtin_amdgcn_ballot_w64((f & 0x000000FFu) != 0), c1 = __builtin_amdgcn_ballot_w64((f & 0x0000FF00u) != 0);
        const uint64_t c2 = __builtin_amdgcn_ballot_w64((f & 0x00FF0000u) != 0), c3 = __builtin_amdgcn_ballot_w64((f & 0xFF000000u) != 0);
        if (cmask && lane == 0) {
            uint64_t* dst = cmask + mbase + nbp * 4 + w;
            dst[0] = c0; dst[cmask_words] = c1; dst[2 * cmask_words] = c2; dst[3 * cmask_words] = c3;
        }
    };
    int nb = 0;
    // the sorted id of a batch's record is fetched one batch ahead: one dependent load less at the head of a batch
    int id_next = g.start + (int)threadIdx.x < g.end ? flat[g.start + threadIdx.x] : 0;
    for (int bs = g.start; bs < g.end; bs += CB, ++nb) {
        // ---- staging, part 1 (registers only): record and cell mask
        const int idx = bs + threadIdx.x;
        const bool have = idx < g.end;
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
        unsigned cm = 0;
        const int64_t id = id_next;
        if (idx + CB < g.end) id_next = flat[idx + CB];
        bool hard = false;
        if (have) {
            ra = splats[id * 3 + 0]; rb = splats[id * 3 + 1]; rc = splats[id * 3 + 2];
            // (same record-level test as the backward's staging, gs_blend.hip: conic safely positive definite, opacity below
            // the clamp; anything else -- NaNs included -- takes the full trip)
            hard = !(ra.z <= 0.998f && ra.w > 0.f && rb.y > 0.f && (ra.w * rb.y - rb.x * rb.x) >= 2e-3f * (ra.w * rb.y));
            cm = no_cull ? 0xFFFFu : cell_mask16(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, g.tx0, g.ty0);
        }
        // every pixel of the tile saturated?  (this barrier also ends the previous batch's trips: sR / sCm are free)
        if (__syncthreads_and(s.thr > 1.0f)) break;
        // ---- staging, part 2: the previous batch's contribution words leave, then the record in q-form (flags cleared)
        // and its cell mask take the slot
        if (nb > 0) flags_out(nb - 1);
        {
            float4* slot = sR + threadIdx.x * (SLOT / 16);
            slot[0] = make_float4(ra.x, ra.y, ra.z, -0.5f * LOG2E * ra.w);
            slot[1] = make_float4(-LOG2E * rb.x, -0.5f * LOG2E * rb.y, rb.z, rb.w);
            slot[2] = make_float4(rc.x, 0.f, 0.f, 0.f);
        }
        sCm[threadIdx.x] = (uint16_t)cm;
        sFlag[threadIdx.x] = 0u;
        if (__builtin_amdgcn_ballot_w64(hard) != 0 && lane == 0) sHard[nb & 1] = 1;
        if (threadIdx.x == 0) sHard[(nb + 1) & 1] = 0;   // (last read before this batch's first barrier)
        __syncthreads();
        // ---- trips: every row walks its own list; the wave runs until its longest list is done
        if (__builtin_amdgcn_ballot_w64(s.thr < 1.0f) != 0) {
            const uint2 lens = build_wave_lists(sCm, lp, g.cell, lane);
            const int l0 = lens.x & 0xFFFF, l1 = lens.x >> 16, l2 = lens.y & 0xFFFF, l3 = lens.y >> 16;
            // Windows of 32 list positions; inside a window four trips per iteration: the four list entries arrive as one
            // 8-byte read issued an iteration ahead, the record reads run two trips ahead of the bodies.
            auto windows = [&](auto lean_tag) {
            constexpr bool LEAN = decltype(lean_tag)::value;
            uint2 e = *reinterpret_cast<const uint2*>(lp);
            for (int k0 = 0;; k0 += 32) {
                // the trips still needed: the longest list among the rows that still have a live pixel (checked once per
                // window: a wave whose pixels have all saturated stops here instead of at the end of the batch --
                // tools/cell_stats.py: 0.745 -> 0.68 of the quadrant trips at SYNTH-1M)
                const uint64_t live = __builtin_amdgcn_ballot_w64(s.thr < 1.0f);
                int need = (live & 0xFFFFull) ? l0 : 0;
                need = max(need, (live & 0xFFFF0000ull) ? l1 : 0);
                need = max(need, (live & 0xFFFF00000000ull) ? l2 : 0);
                need = max(need, (live & 0xFFFF000000000000ull) ? l3 : 0);
                const int n4 = (need + 3) & ~3;
                if (k0 >= n4) break;
                unsigned cbits = 0;
                const int kend = min(k0 + 32, n4);
                unsigned sb = 1u;
                for (int k = k0; k < kend; k += 4, sb <<= 4) {
                    const uint2 en = *reinterpret_cast<const uint2*>(lp + k + 4);   // < CL_CAP: the rows have slack
                    const int o0 = e.x & 0xFFFFu, o1 = e.x >> 16, o2 = e.y & 0xFFFFu, o3 = e.y >> 16;
#define ST3R_LD(o, A, Q, CBV)                                                       \
    const float4 A = *reinterpret_cast<const float4*>(sRb + o);                     \
    const float4 Q = *reinterpret_cast<const float4*>(sRb + o + 16);                \
    const float CBV = *reinterpret_cast<const float*>(sRb + o + 32);
                    // two records in flight: all twelve reads up front cost 12 more registers and a wave per SIMD (measured
                    // slower: 1.14 vs 1.07 ms on the frozen scene)
                    ST3R_LD(o0, a0, q0, c0) ST3R_LD(o1, a1, q1, c1)
                    asm volatile("" ::"v"(c0)); cell_trip<LEAN>(a0, q0, c0, o0, bs, s, cbits, sb);
                    ST3R_LD(o2, a2, q2, c2)
                    asm volatile("" ::"v"(c1)); cell_trip<LEAN>(a1, q1, c1, o1, bs, s, cbits, sb << 1);
                    ST3R_LD(o3, a3, q3, c3)
                    asm volatile("" ::"v"(c2)); cell_trip<LEAN>(a2, q2, c2, o2, bs, s, cbits, sb << 2);
                    asm volatile("" ::"v"(c3)); cell_trip<LEAN>(a3, q3, c3, o3, bs, s, cbits, sb << 3);
#undef ST3R_LD
                    e = en;
                }
                // the window's contribution bits: OR over the row, then lane idx flags the records of positions 2 idx, 2 idx + 1
                if (cmask) {
                    asm volatile("s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                                 : "+v"(cbits));
                    const int p2 = 2 * (lane & 15);
                    const unsigned two = (cbits >> p2) & 3u;
                    if (two) {
                        const unsigned ee = *reinterpret_cast<const unsigned*>(lp + k0 + p2);
                        if (two & 1u) my_flags[4 * (((ee & 0xFFFFu) >> 4) * 43691u >> 17)] = 1;
                        if (two & 2u) my_flags[4 * (((ee >> 16) >> 4) * 43691u >> 17)] = 1;
                    }
                }
            }
            };
            if (__builtin_amdgcn_readfirstlane(sHard[nb & 1]) == 0) windows(std::true_type{}); else windows(std::false_type{});
        }
    }
    // the last batch that ran: its flags are complete once every wave is past its trips
    __syncthreads();
    if (nb > 0) flags_out(nb - 1);
    if (g.inside) {
        const int64_t p = ((int64_t)g.cam * H + g.i) * W + g.j;
        out_rgb[3 * p] = s.r; out_rgb[3 * p + 1] = s.g; out_rgb[3 * p + 2] = s.b;
        out_alpha[p] = 1.0f - s.T;
        last_ids[p] = s.cur == 0x7fffffff ? s.cur : s.cur - 1;
    }
    if (tile_nb && threadIdx.x == 0) tile_nb[g.lb] = nb;
}

int st3r_blend_fwd_cells_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                              const float* splats, const int32_t* offsets, const int32_t* flat, float* rgb, float* alpha,
                              int32_t* last_ids, uint64_t* cmask, int64_t cmask_words, int32_t* tile_nb) {
    const int total = C * tile_w * tile_h;
    hipLaunchKernelGGL(k_blend_fwd_cells, dim3(total), dim3(BLK), 0, s, C, W, H, tile_w, tile_h, (const float4*)splats,
                       offsets, flat, rgb, alpha, last_ids, cmask, cmask_words, tile_nb, ctx->debug_flags & 1);
    LAUNCH_CHECK();
    return ST3R_OK;
}
