// K4: stable ascending LSD radix sort of (key, int32 value) pairs, hand-written for gfx950.
// Replaces cub::DeviceRadixSort::SortPairs as used by gsplat's isect_tiles (starster/gs.py:76); round 1 called
// rocPRIM here.
//
// Structure ("onesweep", Adinets & Merrill 2022, restated for wave64 and 160 KB of LDS):
//   k_rs_hist        one pass over the keys: 256-bin histograms of ALL digits at once (LDS atomics; a wave whose
//                    lanes agree on a digit -- the rule for the high digits of depth / tile keys -- adds once);
//   (the exclusive scan of each histogram = global base of every bin is taken inside k_rs_pass, by every workgroup)
//   k_rs_pass        one launch per 8-bit digit.  A workgroup (16 waves; 8 for small inputs) owns a tile of 16384 (32-bit keys) or 8192
//                    (64-bit keys) consecutive items.  Ranking is wave-local and stable: wave w holds items
//                    [w*64*IPT, (w+1)*64*IPT) of the tile as IPT rows of 64 consecutive items; per row the lanes
//                    find their equals with 8 ballots (match-any), rank = running wave count of the digit (LDS) +
//                    number of equal lanes below.  The per-digit tile counts then travel through a chained scan with
//                    decoupled look-back: one 8-byte {flag, count} word per (tile, digit), published and polled
//                    with relaxed agent-scope atomics (flag and value in ONE word, so no fences are needed and the
//                    protocol is independent of XCD placement); tiles are handed out by an atomic ticket so that
//                    every predecessor of a running tile is itself running.  Items are regrouped by digit in LDS
//                    and leave in runs that are contiguous in the destination.
// Traffic per pass: one read and one write of keys and values (+ 2 KB of status words per tile).
#include "radix_sort.h"

namespace {

constexpr int RB = 8;
constexpr int RADIX = 1 << RB;
// Two tile shapes.  Large inputs: 1024 threads x 16 (32-bit keys) / 8 (64-bit keys) items, one workgroup per CU -- the
// per-tile costs (256 status words, the chained scan) dominate, smaller tiles measured 12-25 % slower (tools/experiments/abl3.sh).
// Small inputs (fewer than one large tile per CU: a rank's share of a view-sharded job): 512 threads x 8 / 4 items, so that
// the tiles still cover the chip.
#ifndef RS_THREADS
#define RS_THREADS 1024
#endif
#ifndef RS_SMALL_BELOW
#define RS_SMALL_BELOW 100   // large tiles needed for the large shape (1 M keys: 0.118 -> 0.098 ms; at 200 tiles the large shape wins)
#endif
constexpr int THREADS_L = RS_THREADS, THREADS_S = 512;
constexpr int HIST_THREADS = 256;
constexpr int HIST_ITEMS = 16;
constexpr int MAX_PASSES = 8;
constexpr int HIST_COPIES = 8;   // global histogram copies (workgroup b adds into copy b % 8; 32 copies measured equal)

typedef unsigned long long u64;
constexpr u64 FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, VALUE_MASK = (1ull << 62) - 1;

template <typename K> struct Traits;
#ifndef RS_IPT32
#define RS_IPT32 16
#endif
template <> struct Traits<uint32_t> { static constexpr int IPT = RS_IPT32, IPT_S = 8; };
template <> struct Traits<uint64_t> { static constexpr int IPT = 8, IPT_S = 4; };

__device__ __forceinline__ int wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// value of lane q of this lane's quad (DPP quad_perm: one VALU move, no LDS)
// value of lane q of this lane's PAIR (lanes 2k, 2k+1)
__device__ __forceinline__ uint32_t pair_lane_u32(uint32_t v, int q) {
    return q == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xf, 0xf, true)
                  : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t quad_lane_u32(uint32_t v, int q) {
    switch (q) {
        case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true);
        case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true);
        case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, true);
        default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, true);
    }
}

template <typename K>
__global__ __launch_bounds__(HIST_THREADS) void k_rs_hist(const K* __restrict__ keys, int64_t n, int begin_bit,
                                                          int end_bit, int passes, uint32_t* __restrict__ hist,
                                                          const int32_t* __restrict__ n_dev) {
    // one private copy of the histograms per wave: LDS atomics of different waves never meet on an address
    __shared__ uint32_t sh[HIST_THREADS / 64][MAX_PASSES * RADIX];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));   // the item count lives on the device; `n` is the launch capacity
    for (int i = threadIdx.x; i < (HIST_THREADS / 64) * MAX_PASSES * RADIX; i += HIST_THREADS) (&sh[0][0])[i] = 0;
    __syncthreads();
    uint32_t* mine = sh[threadIdx.x >> 6];
    constexpr int CH = HIST_THREADS * HIST_ITEMS;
    for (int64_t base = (int64_t)blockIdx.x * CH; base < n; base += (int64_t)gridDim.x * CH) {
        K k[HIST_ITEMS];
#pragma unroll
        for (int i = 0; i < HIST_ITEMS; ++i) {   // all loads of the chunk in flight before the first atomic
            const int64_t idx = base + i * HIST_THREADS + threadIdx.x;
            k[i] = idx < n ? keys[idx] : K(0);
        }
#pragma unroll
        for (int i = 0; i < HIST_ITEMS; ++i) {
            if (base + i * HIST_THREADS + threadIdx.x < n) {
                for (int p = 0; p < passes; ++p) {
                    const int shift = begin_bit + RB * p;
                    const int bits = min(RB, end_bit - shift);
                    atomicAdd(&mine[p * RADIX + ((uint32_t)(k[i] >> shift) & ((1u << bits) - 1u))], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * RADIX; i += HIST_THREADS) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < HIST_THREADS / 64; ++w) t += sh[w][i];
        if (t) atomicAdd(&hist[(blockIdx.x & (HIST_COPIES - 1)) * (MAX_PASSES * RADIX) + i], t);   // copy b % HIST_COPIES: fewer adds meet on an address
    }
}

// Segmented, biased variant of the histogram (level-1 sort of the fused training calls): segment g = blockIdx.y holds the
// items [g seg_n, (g + 1) seg_n); a key is first mapped to key - krange[0] (the sentinel 0x1FFFFFFF of a culled pair to
// krange[1]), and every segment gets its own histograms of the four digits.
__device__ __forceinline__ uint32_t seg_key(uint32_t raw, uint32_t kmin, uint32_t ktop) {
    return raw == 0x1FFFFFFFu ? ktop : raw - kmin;
}
__global__ __launch_bounds__(HIST_THREADS) void k_rs_hist_seg(const uint32_t* __restrict__ keys, int64_t seg_n,
                                                              uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ krange) {
    __shared__ uint32_t sh[HIST_THREADS / 64][4 * RADIX];
    const uint32_t kmin = krange[0], ktop = krange[1];
    const int passes = (int)krange[2];
    for (int i = threadIdx.x; i < (HIST_THREADS / 64) * 4 * RADIX; i += HIST_THREADS) (&sh[0][0])[i] = 0;
    __syncthreads();
    uint32_t* mine = sh[threadIdx.x >> 6];
    const uint32_t* kseg = keys + (int64_t)blockIdx.y * seg_n;
    constexpr int CH = HIST_THREADS * HIST_ITEMS;
    for (int64_t base = (int64_t)blockIdx.x * CH; base < seg_n; base += (int64_t)gridDim.x * CH) {
        uint32_t k[HIST_ITEMS];
#pragma unroll
        for (int i = 0; i < HIST_ITEMS; ++i) {
            const int64_t idx = base + i * HIST_THREADS + threadIdx.x;
            k[i] = idx < seg_n ? kseg[idx] : 0u;
        }
#pragma unroll
        for (int i = 0; i < HIST_ITEMS; ++i) {
            if (base + i * HIST_THREADS + threadIdx.x < seg_n) {
                const uint32_t kk = seg_key(k[i], kmin, ktop);
                for (int p = 0; p < passes; ++p) atomicAdd(&mine[p * RADIX + ((kk >> (RB * p)) & (RADIX - 1))], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * RADIX; i += HIST_THREADS) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < HIST_THREADS / 64; ++w) t += sh[w][i];
        if (t) atomicAdd(&hist[blockIdx.y * (MAX_PASSES * RADIX) + i], t);
    }
}

// SEG (level-1 sort of the fused training calls; K = uint32_t): the input is n_seg segments of seg_n items (one per camera),
// each sorted on its own -- tile t serves segment t / tiles_per_seg, the chained scan restarts at a segment's first tile --,
// the keys are biased on the way in (pass 0: seg_key), and the NUMBER of passes is read from device memory (krange[2], set
// by the projection's reduction from the depth range it saw: three when the biased keys stay below 2^24, else four): launch
// `ps` returns at once when ps >= passes, and source / destination of a pass follow from the pass count so that the last
// pass that runs writes kout / vout (kin = the caller's input, ktmp / vtmp = scratch).
template <typename K, int THREADS, int IPT, bool SEG = false>
__global__ __launch_bounds__(THREADS) void k_rs_pass(const K* __restrict__ kin, const int32_t* __restrict__ vin,
                                                     K* __restrict__ kout, int32_t* __restrict__ vout, int64_t n,
                                                     int shift, int bits, const uint32_t* __restrict__ hist_base,
                                                     u64* status, uint32_t* tile_counter,
                                                     const int32_t* __restrict__ n_dev, int64_t seg_n = 0,
                                                     int tiles_per_seg = 0, const uint32_t* __restrict__ krange = nullptr,
                                                     int ps = 0, K* __restrict__ ktmp = nullptr,
                                                     int32_t* __restrict__ vtmp = nullptr) {
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));   // count on the device, grid sized for the capacity `n`
    constexpr int TILE = THREADS * IPT, WAVES = THREADS / 64;
    uint32_t kmin = 0, ktop = 0;
    bool last_seg_pass = false;
    if constexpr (SEG) {
        const int np = (int)krange[2];
        last_seg_pass = ps == np - 1;
        if (ps >= np) return;                                   // (uniform over the launch)
        const bool to_out = ((np - 1 - ps) & 1) == 0;           // the last pass that runs writes the caller's output
        const K* src_k = ps == 0 ? kin : (to_out ? ktmp : kout);
        const int32_t* src_v = ps == 0 ? vin : (to_out ? vtmp : vout);
        K* dst_k = to_out ? kout : ktmp;
        int32_t* dst_v = to_out ? vout : vtmp;
        kin = src_k; vin = src_v; kout = dst_k; vout = dst_v;
        kmin = krange[0]; ktop = krange[1];
        shift = RB * ps; bits = RB;
    }
    __shared__ K sbuf[TILE];
    __shared__ int32_t svals[TILE];   // values regroup together with the keys: one trip through LDS, one store phase
    __shared__ uint32_t whist[WAVES][RADIX];
    __shared__ uint32_t lbase[RADIX];
    __shared__ uint32_t tcnt[RADIX];
    __shared__ u64 gexc[RADIX];
    __shared__ long long gofs[RADIX];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t hsum[4];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    uint32_t first_tile = 0;      // first tile of this tile's chain (SEG: of its segment)
    int64_t tbase, seg_base = 0;
    int tcount;
    if constexpr (SEG) {
        const uint32_t sg = tile / (uint32_t)tiles_per_seg;
        first_tile = sg * (uint32_t)tiles_per_seg;
        seg_base = (int64_t)sg * seg_n;
        const int64_t in_seg = (int64_t)(tile - first_tile) * TILE;
        tbase = seg_base + in_seg;
        tcount = (int)min((int64_t)TILE, seg_n - in_seg);
        hist_base += (size_t)sg * (MAX_PASSES * RADIX);
    } else {
        if ((int64_t)tile * TILE >= n) return;   // (capacity launch) a ticket past the last tile: uniform over the workgroup
        tbase = (int64_t)tile * TILE;
        tcount = (int)min((int64_t)TILE, n - tbase);
    }
    const uint32_t dmask = (1u << bits) - 1u;
    const int wbase = w * 64 * IPT + lane;

    K key[IPT];
    int32_t val[IPT];
    uint32_t rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int li = wbase + i * 64;
        key[i] = li < tcount ? kin[tbase + li] : K(0);
        if constexpr (SEG) { if (ps == 0) key[i] = (K)seg_key((uint32_t)key[i], kmin, ktop); }
    }
    // this pass's global digit counts (the HIST_COPIES partial histograms of k_rs_hist; in flight while the keys are ranked).
    // Their exclusive scan -- the global base of every bin -- is taken by every workgroup for itself further down, next to
    // the tile-local one (round 5: it used to be a launch of one workgroup, k_rs_scan_hist, between the histogram and the
    // first pass: 4.6 us twice per step).
    uint32_t hv = 0;
    if (tid < RADIX) {
        if constexpr (SEG) hv = hist_base[tid];   // (one histogram per segment)
        else {
#pragma unroll
            for (int c = 0; c < HIST_COPIES; ++c) hv += hist_base[c * (MAX_PASSES * RADIX) + tid];
        }
    }
    // SEG, pass 0: the values ARE the item indices (pair ids) -- nothing is read, and the projection does not write them
    const bool iota_vals = SEG && ps == 0;
    const bool has_vals = SEG || vin != nullptr;
    if (iota_vals) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) val[i] = (int32_t)(tbase + wbase + i * 64);
    } else if (vin) {   // in flight while the keys are ranked
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int li = wbase + i * 64;
            val[i] = li < tcount ? vin[tbase + li] : 0;
        }
    }
    const u64 lt = (1ull << lane) - 1ull;
    uint32_t* wh = whist[w];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const bool valid = wbase + i * 64 < tcount;
        const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t below = (uint32_t)__popcll(m & lt);
        const uint32_t prev = wh[d];
        if (valid && below == 0) wh[d] = prev + (uint32_t)__popcll(m);
        rank[i] = prev + below;
    }
    __syncthreads();

    // per digit: exclusive scan over the waves (in place), tile count -> published as this tile's aggregate
    uint32_t cnt = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            const uint32_t c = whist[ww][tid];
            whist[ww][tid] = cnt;
            cnt += c;
        }
        if (tile != first_tile)
            __hip_atomic_store(status + (size_t)tile * RADIX + tid, FLAG_AGG | (u64)cnt, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        tcnt[tid] = cnt;
    }
    // tile-local exclusive scan of the digit counts
    const uint32_t inc = wave_incl_scan_u32(cnt, lane);
    const uint32_t hinc = wave_incl_scan_u32(hv, lane);
    if (tid < RADIX && lane == 63) { wsum[w] = inc; hsum[w] = hinc; }
    __syncthreads();
    uint32_t hexc = 0;   // global base of bin `tid` (threads below RADIX)
    if (tid < RADIX) {
        uint32_t base = 0, hb = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { base += i < w ? wsum[i] : 0u; hb += i < w ? hsum[i] : 0u; }
        lbase[tid] = base + inc - cnt;
        hexc = hb + hinc - hv;
    }
    // chained scan over the tiles, LB lanes per digit: lane j of a digit's group reads the status word of the
    // (j+1)-th predecessor, then the next LB ...; the group adds aggregates up to the nearest inclusive prefix.
    // Every thread of the workgroup takes part (RADIX * LB == THREADS), so one trip of a few hundred ns covers LB
    // predecessors instead of one.
    {
        constexpr int LB = THREADS / RADIX;
        static_assert(LB == 4 || LB == 2, "the look-back group is a DPP quad or pair");
        const int d = tid / LB, j = tid % LB;
        u64 excl = 0;
        if (tile != first_tile) {
            bool done = false;
            for (int64_t t0 = (int64_t)tile - 1; !done; t0 -= LB) {
                const int64_t t = t0 - j;
                u64 v = FLAG_INC;   // before the first tile: an inclusive prefix of zero
                if (t >= (int64_t)first_tile) {
                    const u64* pt = status + (size_t)t * RADIX + d;
                    v = __hip_atomic_load(pt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    while ((v >> 62) == 0) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(pt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // a predecessor always holds an earlier ticket, i.e. it is running: this bound (seconds) can
                        // only trip on a broken device or a protocol bug, and then it must be loud rather than a hang
                        if (++spins > (1u << 26)) __builtin_trap();
                    }
                }
                const uint32_t val = (uint32_t)(v & VALUE_MASK), isinc = (uint32_t)(v >> 62) == 2u;
#pragma unroll
                for (int q = 0; q < LB; ++q) {   // nearest predecessor first
                    const uint32_t vq = LB == 4 ? quad_lane_u32(val, q) : pair_lane_u32(val, q);
                    const uint32_t iq = LB == 4 ? quad_lane_u32(isinc, q) : pair_lane_u32(isinc, q);
                    if (!done) { excl += vq; done = iq != 0; }
                }
            }
        }
        if (j == 0) {
            __hip_atomic_store(status + (size_t)tile * RADIX + d, FLAG_INC | (excl + tcnt[d]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            gexc[d] = excl;
        }
    }
    __syncthreads();
    if (tid < RADIX) gofs[tid] = (long long)((u64)hexc + gexc[tid]) - (long long)lbase[tid] + seg_base;
    __syncthreads();

    // regroup by digit in LDS, then leave in runs that are contiguous in the destination
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (wbase + i * 64 < tcount) {
            const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
            const uint32_t pos = lbase[d] + wh[d] + rank[i];
            sbuf[pos] = key[i];
            if (has_vals) svals[pos] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int p = i * THREADS + tid;
        if (p < tcount) {
            const K k = sbuf[p];
            const uint32_t d = (uint32_t)(k >> shift) & dmask;
            const long long o = gofs[d] + p;
            if (!(SEG && last_seg_pass)) kout[o] = k;   // (nobody reads the level-1 keys once the pairs are in order)
            if (has_vals) vout[o] = svals[p];
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename K>
int sort_pairs(st3r_ctx* ctx, hipStream_t s, int64_t n, int begin_bit, int end_bit, const K* keys_in,
               const int32_t* vals_in, K* keys_out, int32_t* vals_out, const int32_t* n_dev = nullptr) {
    if (n == 0) return ST3R_OK;
    if (end_bit > (int)sizeof(K) * 8) end_bit = (int)sizeof(K) * 8;
    if (begin_bit < 0 || begin_bit >= end_bit) { st3r_set_error("radix sort: empty bit range"); return ST3R_ERR_INVALID; }
    const int passes = (end_bit - begin_bit + RB - 1) / RB;
    constexpr int TILE_L = THREADS_L * Traits<K>::IPT, TILE_S = THREADS_S * Traits<K>::IPT_S;
    const bool small = (n + TILE_L - 1) / TILE_L < RS_SMALL_BELOW;   // (with a device count: decided on the capacity)
    const int64_t TILE = small ? TILE_S : TILE_L;
    const int64_t ntiles = (n + TILE - 1) / TILE;
    const size_t hist_bytes = align256(sizeof(uint32_t) * (size_t)(HIST_COPIES * MAX_PASSES * RADIX + MAX_PASSES));
    const size_t status_bytes = sizeof(u64) * (size_t)passes * (size_t)ntiles * RADIX;
    const size_t meta_bytes = hist_bytes + align256(status_bytes);
    const bool need_tmp = passes > 1;
    const size_t tk_bytes = need_tmp ? align256(sizeof(K) * (size_t)n) : 0;
    const size_t tv_bytes = need_tmp && vals_in ? align256(sizeof(int32_t) * (size_t)n) : 0;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, meta_bytes + tk_bytes + tv_bytes, &p);
    if (rc) return rc;
    char* base = (char*)p;
    uint32_t* hist = (uint32_t*)base;
    uint32_t* counters = hist + HIST_COPIES * MAX_PASSES * RADIX;
    u64* status = (u64*)(base + hist_bytes);
    K* tk = (K*)(base + meta_bytes);
    int32_t* tv = (int32_t*)(base + meta_bytes + tk_bytes);
    HIP_TRY(hipMemsetAsync(base, 0, meta_bytes, s));
    const int hist_blocks = (int)min((int64_t)1024, (n + HIST_THREADS * HIST_ITEMS - 1) / (HIST_THREADS * HIST_ITEMS));
    hipLaunchKernelGGL(k_rs_hist<K>, dim3(hist_blocks), dim3(HIST_THREADS), 0, s, keys_in, n, begin_bit, end_bit, passes,
                       hist, n_dev);
    const K* kin = keys_in;
    const int32_t* vin = vals_in;
    for (int ps = 0; ps < passes; ++ps) {
        // the last pass writes the caller's output; before that the passes alternate between it and the scratch
        const bool to_out = ((passes - 1 - ps) & 1) == 0;
        K* ko = to_out ? keys_out : tk;
        int32_t* vo = to_out ? vals_out : tv;
        const int shift = begin_bit + RB * ps;
        const int bits = min(RB, end_bit - shift);
        if (small)
            hipLaunchKernelGGL((k_rs_pass<K, THREADS_S, Traits<K>::IPT_S>), dim3((unsigned)ntiles), dim3(THREADS_S), 0, s, kin,
                               vin, ko, vin ? vo : nullptr, n, shift, bits, hist + ps * RADIX,
                               status + (size_t)ps * ntiles * RADIX, counters + ps, n_dev);
        else
            hipLaunchKernelGGL((k_rs_pass<K, THREADS_L, Traits<K>::IPT>), dim3((unsigned)ntiles), dim3(THREADS_L), 0, s, kin,
                               vin, ko, vin ? vo : nullptr, n, shift, bits, hist + ps * RADIX,
                               status + (size_t)ps * ntiles * RADIX, counters + ps, n_dev);
        kin = ko;
        if (vin) vin = vo;
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}

// n_seg segments of seg_n (key, value) pairs each, every segment sorted on its own by seg_key(key) (see k_rs_pass<.., SEG>);
// krange (device): bias, sentinel key and pass count
int sort_pairs_seg(st3r_ctx* ctx, hipStream_t s, int64_t seg_n, int n_seg, const uint32_t* keys_in, const int32_t* vals_in,
                   uint32_t* keys_out, int32_t* vals_out, const uint32_t* krange) {
    if (seg_n == 0 || n_seg == 0) return ST3R_OK;
    typedef uint32_t K;
    constexpr int PASSES = 4;   // launches; the passes that run: krange[2]
    constexpr int TILE_L = THREADS_L * Traits<K>::IPT, TILE_S = THREADS_S * Traits<K>::IPT_S;
    const int64_t n = seg_n * n_seg;
    const bool small = (n + TILE_L - 1) / TILE_L < RS_SMALL_BELOW;
    const int64_t TILE = small ? TILE_S : TILE_L;
    const int tiles_per_seg = (int)((seg_n + TILE - 1) / TILE);
    const int64_t ntiles = (int64_t)tiles_per_seg * n_seg;
    const size_t hist_bytes = align256(sizeof(uint32_t) * ((size_t)n_seg * MAX_PASSES * RADIX + MAX_PASSES));
    const size_t status_bytes = sizeof(u64) * (size_t)PASSES * (size_t)ntiles * RADIX;
    const size_t meta_bytes = hist_bytes + align256(status_bytes);
    const size_t tk_bytes = align256(sizeof(K) * (size_t)n), tv_bytes = align256(sizeof(int32_t) * (size_t)n);
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, meta_bytes + tk_bytes + tv_bytes, &p);
    if (rc) return rc;
    char* base = (char*)p;
    uint32_t* hist = (uint32_t*)base;
    uint32_t* counters = hist + (size_t)n_seg * MAX_PASSES * RADIX;
    u64* status = (u64*)(base + hist_bytes);
    K* tk = (K*)(base + meta_bytes);
    int32_t* tv = (int32_t*)(base + meta_bytes + tk_bytes);
    HIP_TRY(hipMemsetAsync(base, 0, meta_bytes, s));
    const int hist_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024 / n_seg, (seg_n + HIST_THREADS * HIST_ITEMS - 1) / (HIST_THREADS * HIST_ITEMS)));
    hipLaunchKernelGGL(k_rs_hist_seg, dim3(hist_blocks, n_seg), dim3(HIST_THREADS), 0, s, keys_in, seg_n, hist, krange);
    for (int ps = 0; ps < PASSES; ++ps) {
        if (small)
            hipLaunchKernelGGL((k_rs_pass<K, THREADS_S, Traits<K>::IPT_S, true>), dim3((unsigned)ntiles), dim3(THREADS_S), 0, s,
                               keys_in, vals_in, keys_out, vals_out, n, 0, RB, hist + ps * RADIX,
                               status + (size_t)ps * ntiles * RADIX, counters + ps, (const int32_t*)nullptr, seg_n,
                               tiles_per_seg, krange, ps, tk, tv);
        else
            hipLaunchKernelGGL((k_rs_pass<K, THREADS_L, Traits<K>::IPT, true>), dim3((unsigned)ntiles), dim3(THREADS_L), 0, s,
                               keys_in, vals_in, keys_out, vals_out, n, 0, RB, hist + ps * RADIX,
                               status + (size_t)ps * ntiles * RADIX, counters + ps, (const int32_t*)nullptr, seg_n,
                               tiles_per_seg, krange, ps, tk, tv);
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}

}  // namespace

int st3r_radix_sort_u32_segments(st3r_ctx* ctx, hipStream_t s, int64_t seg_n, int n_seg, const uint32_t* keys_in,
                                 const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out, const uint32_t* krange) {
    return sort_pairs_seg(ctx, s, seg_n, n_seg, keys_in, vals_in, keys_out, vals_out, krange);
}

int st3r_radix_sort_u32(st3r_ctx* ctx, hipStream_t s, int64_t n, int begin_bit, int end_bit, const uint32_t* keys_in,
                        const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out) {
    return sort_pairs<uint32_t>(ctx, s, n, begin_bit, end_bit, keys_in, vals_in, keys_out, vals_out);
}

int st3r_radix_sort_u64(st3r_ctx* ctx, hipStream_t s, int64_t n, int begin_bit, int end_bit, const uint64_t* keys_in,
                        const int32_t* vals_in, uint64_t* keys_out, int32_t* vals_out) {
    return sort_pairs<uint64_t>(ctx, s, n, begin_bit, end_bit, keys_in, vals_in, keys_out, vals_out);
}

// the same with the item count in device memory: `n_cap` sizes scratch and grids, the kernels sort the first
// min(*n_dev, n_cap) items (no host round trip between the kernel that counts the items and the sort)
int st3r_radix_sort_u32_devcount(st3r_ctx* ctx, hipStream_t s, int64_t n_cap, const int32_t* n_dev, int begin_bit,
                                 int end_bit, const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out,
                                 int32_t* vals_out) {
    return sort_pairs<uint32_t>(ctx, s, n_cap, begin_bit, end_bit, keys_in, vals_in, keys_out, vals_out, n_dev);
}
