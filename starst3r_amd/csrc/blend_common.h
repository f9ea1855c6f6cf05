// Device helpers shared by the two blend translation units (gs_blend.hip: quadrant formulation, stand-alone entry
// points; gs_blend_cells.hip: 4x4-cell lists, the fused training path).
#pragma once
#include "common.h"
#include "tile_rect.h"

#define BLK 256
#define LOG2E 1.4426950408889634f

// Workgroup b runs on XCD b % 8 (each XCD has its own L2).  Tiles are handed out in groups of G consecutive
// tiles per XCD, the groups round-robin over the XCDs: neighbouring tiles (which share Gaussians) meet in one L2.
// A group is one TILE ROW (XCD_GROUP), whatever the number of views: every XCD then gets every eighth row of every camera --
// the dense image centre and the sparse borders are spread over all XCDs.  Rounds 2-5 gave a whole camera to an XCD when the
// views were a multiple of 8; measured in round 6 (alternating same-box A/B, frozen SYNTH-1M scene): rows instead of
// cameras -0.035 +- 0.006 ms on the blend backward (the cameras' record counts differ by a few per cent and the kernel
// ends with its slowest XCD), two rows per group equal, half rows +0.03, quarter rows +1.6 ms (vertical stripes: the XCDs
// that own the centre columns do most of the work), 17 rows +0.16.
#ifndef XCD_GROUP
#define XCD_GROUP(C, n_tiles, tile_w) (tile_w)
#endif
__device__ __forceinline__ int xcd_remap(int bid, int total, int G) {
    const int n_full = (total / (8 * G)) * (8 * G);
    if (bid >= n_full) return bid;
    const int xcd = bid & 7, k = bid >> 3;
    const int round = k / G, within = k - round * G;
    return (round * 8 + xcd) * G + within;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Stage record `id` into LDS slot t (q-form) and return the 4-bit quadrant relevance.
// q-form of a record in LDS slot t (see the arithmetic note above); forward and backward stage through this one function
// LDS record: 3 x float4 = (x y opacity qa | qb qc r g | b - - -); one address register serves the three reads
__device__ __forceinline__ void stage_qform(const float4& a, const float4& b, const float4& c, int t, float4* sR) {
    sR[3 * t + 0] = make_float4(a.x, a.y, a.z, -0.5f * LOG2E * a.w);
    sR[3 * t + 1] = make_float4(-LOG2E * b.x, -0.5f * LOG2E * b.y, b.z, b.w);
    sR[3 * t + 2] = make_float4(c.x, 0.f, 0.f, 0.f);
}

// word index of the first 64-record chunk of tile lb in the contribution-mask arrays
// (a tile of `len` records uses 4*ceil(len/256) <= floor(len/64) + 4 words, hence the 4*lb slack)
__device__ __forceinline__ int64_t mask_base(int lb, int start) { return (int64_t)(start >> 6) + 4 * (int64_t)lb; }

// The exponent of a record at a pixel, P = -log2(e) sigma = dx (qa dx + qb dy) + qc dy^2, with ONE fixed association
// and contraction: every blend kernel (forward and backward, quadrant and cell formulation) must take identical
// include / skip decisions, and hipcc contracts the plain expression differently from kernel to kernel (round 3: the
// quadrant forward computed fma(dx, lx, (qc dy) dy), the backward fma(dy, qc dy, dx lx)).
__device__ __forceinline__ float blend_power(float dx, float dy, float qa, float qb, float qc) {
#pragma clang fp contract(off)
    const float lx = __builtin_fmaf(dx, qa, qb * dy);
    return __builtin_fmaf(dx, lx, (qc * dy) * dy);
}

// Compare into a scalar register pair / select under such a mask, spelled out: hipcc keeps a compare result that must
// outlive the next compare in a VGPR (v_cndmask 0/1 + v_cmp_ne to get the ballot back: two extra VALU instructions per
// trip of the forward loop).  "s_nop 1": wait states between a scalar write of the mask and its use by v_cndmask.
__device__ __forceinline__ uint64_t mask_not_less(float a, float b) {   // lanes with !(a < b)
    uint64_t m;
    asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint64_t mask_not_positive(float a) {        // lanes with !(a > 0)
    uint64_t m;
    asm("v_cmp_nlt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
    return m;
}
__device__ __forceinline__ float zero_unless(uint64_t m, float x) {     // m ? x : 0
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

// reduce9_rows: sums g[0..8] over the 16 lanes of every DPP row (row = one record of the chunk) with DPP adds only:
//   level 1 (lanes l, l ^ 8)   two values per register -- lanes 0-7 keep the first, lanes 8-15 the second (the writes are
//                              restricted with bank_mask; a bank = four consecutive lanes), 9 instructions -> 5 registers;
//   level 2 (lanes l, l ^ 4)   the same trick with row_shl:4 / row_shr:4: 5 instructions -> 3 registers, every bank now
//                              holds a different value;
//   levels 3, 4                inside the quads (quad_perm), all four lanes end with the total: 6 instructions.
// 20 DPP adds (1.7 ns each on this chip) instead of 8 lane swaps (3.4 ns) + 8 adds + 6 DPP adds of reduce9.
// On return, in bank b (lanes 4b .. 4b+3) of every row: k0 = total of g[{0,2,1,3}[b]], k1 = total of g[{4,6,5,7}[b]],
// k2 (bank 0 only) = total of g[8].
__device__ __forceinline__ void reduce9_rows(float g0, float g1, float g2, float g3, float g4, float g5, float g6,
                                             float g7, float g8, float& k0, float& k1, float& k2) {
    float t0, t1, t2, t3, t4;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
        : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "v"(g6), "v"(g7), "v"(g8));
    float u0, u1, u2;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x1\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(u0), "=&v"(u1), "=&v"(u2)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4));
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(u0), "+v"(u1), "+v"(u2));
    k0 = u0; k1 = u1; k2 = u2;
}
#define ACC_VALS 9      // S_x S_y S_o S_xx S_xy S_yy S_r S_g S_b per staged record and wave
#define VT_STRIDE 10    // per-(record, tile) slot: 9 partial gradients + the stamp = 5 x 8 B
#ifndef HB
#define HB 64           // records staged per backward round (a fraction of a forward batch of 256)
#endif
#define HB_WORDS (HB / 64)   // 64-record mask words per round
static_assert(HB == 64, "the backward walks one mask word per round");
#ifndef CHUNK
#define CHUNK 4         // records per transposition chunk (4 or 8); a lane of phase 2 owns CHUNK pixels of one row
#endif
// row layout of phase 2 (lane = part + 16 x record): a record row is 32 + 1 + 32 + 1 float2 -- pixel p sits at p + (p >> 5),
// rows 66 apart -- so that the 32 lanes of a read group (two records x 16 parts, 32-byte stride inside a row) meet in
// 32 different bank pairs
#define PAIR_STRIDE 66
#define PAIR_AT(pix) ((pix) + ((pix) >> 5))

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave complete in order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
