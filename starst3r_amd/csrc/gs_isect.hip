// K2/K3/K5: tile-intersection bookkeeping.
//   scan    -- inclusive prefix sum of tiles_per_gauss (gsplat: torch.cumsum between the two
//              isect_tiles passes): one single-pass kernel with decoupled look-back (k_scan_chained);
//   emit    -- gsplat isect_tiles pass 2: one (key,value) per touched tile (k_isect_emit: the stage API's
//              64-bit keys in pair order; k_isect_gather / k_isect_wg_scan / k_isect_emit_d: the fused
//              path's 32-bit (camera, tile) keys in (camera, depth) order);
//   offsets -- gsplat isect_offset_encode.
// All integer work: results are bit-exact against oracle/gs_oracle.c.
#include "common.h"
#include "tile_rect.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// block-wide inclusive scan of one int per thread (256 threads = 4 waves); returns inclusive value,
// *total receives the block total
__device__ __forceinline__ int block_incl_scan(int v, int* total) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? wsum[i] : 0;
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return inc + base;
}

// Single-pass prefix sum (decoupled look-back, Merrill & Garland 2016, restated for wave64): a workgroup owns a tile of
// 4096 consecutive elements, publishes the tile's sum as soon as it is known, and adds up the published sums of its
// predecessors -- 256 status words per trip, one per thread (chain_lookback256) -- back to the nearest tile whose
// inclusive prefix is already known.  Round 3 ran three launches per scan (reduce, one workgroup over the tile sums,
// downsweep) and read the input twice.
//   status word = flag (2 bits: 1 = tile sum, 2 = inclusive prefix) | launch generation (14 bits) | value (48 bits)
// in ONE 64-bit word published and polled with relaxed agent-scope atomics: no fences, independent of XCD placement.
// The generation makes words of earlier launches invisible, so the status array is never cleared between launches
// (the host clears it when the generation wraps); tiles are handed out by a ticket counter, so every predecessor of a
// running tile is itself running.  Two counters take turns (launch parity = generation & 1): a launch draws from one and
// its first tile zeroes the other for the launch behind it -- no host-side mirror of device state, no clearing launch.
//   RECTS  the input is the packed tile rectangle of every element (x0 | y0 << 16 | w << 32 | h << 48), its count w * h
//   PACK   also leave, per element, exclusive prefix << 32 | x0 | y0 << 10 | w << 20 -- the one word the blend backward
//          gathers per staged record (slot base + rectangle)
// Thread t owns SCAN_ITEMS consecutive elements so the scan order is the array order; the tile travels through LDS so
// that both the loads and the stores are coalesced (padded by one word per 32: the 16-word runs of neighbouring
// threads would otherwise meet in the same banks).  `in` may alias `out` (a tile is loaded before any of it is stored).
typedef unsigned long long u64;
#define SCAN_PAD(i) ((i) + ((i) >> 5))
#define SC_VALUE_MASK 0xFFFFFFFFFFFFull
__device__ __forceinline__ u64 sc_word(u64 flag, uint32_t gen, u64 v) { return (flag << 62) | ((u64)gen << 48) | (v & SC_VALUE_MASK); }

// Look-back of a chained scan by the whole workgroup: thread j polls the status word of the (j+1)-th predecessor of tile
// `k` (k >= 1) of the chain whose words start at `st`, 256 predecessors per trip; the sums of the published tile totals
// back to the nearest predecessor whose inclusive prefix is known give this tile's exclusive prefix (returned to every
// thread).  One wave looking back 64 tiles per trip was measured first: when all tiles of a launch are resident at once
// nobody has an inclusive prefix early, so tile k needs k / 64 trips of a few microseconds each -- the pair-order scan's
// 1954 tiles spent 60 us that way.
__device__ __forceinline__ u64 chain_lookback256(const u64* st, int k, uint32_t gen, u64* s_part, int* s_near) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    u64 excl = 0;
    for (int t0 = k - 1;; t0 -= 256) {
        const int tt = t0 - t;
        u64 wd = sc_word(2, gen, 0);   // in front of the chain's first tile: an inclusive prefix of zero
        if (tt >= 0) {
            unsigned spins = 0;
            for (;;) {
                wd = __hip_atomic_load(st + tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((wd >> 62) != 0 && (uint32_t)((wd >> 48) & 0x3FFF) == gen) break;
                __builtin_amdgcn_s_sleep(1);
                // a predecessor holds an earlier ticket, i.e. it is running: this bound (seconds) can only trip on a
                // broken device or a protocol bug, and then it must be loud rather than a hang
                if (++spins > (1u << 26)) __builtin_trap();
            }
        }
        const u64 incm = __ballot((wd >> 62) == 2);
        const int nearest = incm ? __builtin_ctzll(incm) : 64;   // nearest predecessor of this wave's 64 with a prefix
        u64 val = lane <= nearest ? (wd & SC_VALUE_MASK) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
        if (lane == 0) { s_part[w] = val; s_near[w] = nearest; }
        __syncthreads();
        bool done = false;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww)
            if (!done) { excl += s_part[ww]; done = s_near[ww] < 64; }
        __syncthreads();
        if (done) break;
    }
    return excl;
}

template <bool RECTS, bool PACK>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chained(const int32_t* in, const void* __restrict__ rects,
                                                               int rect32, int64_t n, int32_t* out,
                                                               uint64_t* __restrict__ pack_out, u64* status,
                                                               uint32_t* ticket, uint32_t gen,
                                                               int32_t* __restrict__ total_out,
                                                               int32_t* __restrict__ total_copy,
                                                               int32_t* __restrict__ total_host) {
    __shared__ int tile[SCAN_PAD(SCAN_TILE) + 1];
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) {
        s_tile = atomicAdd(ticket + (gen & 1u), 1u);
        // (the other counter belongs to the previous launch -- finished -- and to the next one -- not started)
        if (s_tile == 0) __hip_atomic_store(ticket + ((gen & 1u) ^ 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t tl = s_tile;
    const int64_t base0 = (int64_t)tl * SCAN_TILE;
    if (base0 >= n) return;   // uniform over the workgroup
    uint32_t geo[PACK ? SCAN_ITEMS : 1];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int e = i * SCAN_THREADS + threadIdx.x;
        const int64_t idx = base0 + e;
        int v = 0;
        if (RECTS) {
            uint32_t org = 0, rw = 0, rh = 0;
            if (idx < n) rect_load(rects, rect32, idx, &org, &rw, &rh);
            v = (int)(rw * rh);
            if (PACK) geo[i] = (org & 0x3FF) | (((org >> 16) & 0x3FF) << 10) | ((rw & 0x3FF) << 20);
        } else if (idx < n) v = in[idx];
        tile[SCAN_PAD(e)] = v;
    }
    __syncthreads();
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = tile[SCAN_PAD(threadIdx.x * SCAN_ITEMS + i)];
        s += v[i];
    }
    int total;
    const int inc = block_incl_scan(s, &total);
    // (a tile total < 2^32: 4096 counts < 2^20 each)
    const u64 tot = (u64)(uint32_t)total;
    // publish the tile's total, add up the predecessors, publish the inclusive prefix
    __shared__ u64 s_part[4];
    __shared__ int s_near[4];
    u64 excl = 0;
    if (tl == 0) {
        if (threadIdx.x == 0) __hip_atomic_store(status, sc_word(2, gen, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (threadIdx.x == 0) __hip_atomic_store(status + tl, sc_word(1, gen, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        excl = chain_lookback256(status, (int)tl, gen, s_part, s_near);
        if (threadIdx.x == 0)
            __hip_atomic_store(status + tl, sc_word(2, gen, excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && base0 + SCAN_TILE >= n) {   // the last tile: the grand total (past 2^31 - 1 it is reported as -1
        const u64 all = excl + tot;                     // -- the callers turn a negative count into an error -- instead of
        total_out[0] = all > 2147483647ull ? -1 : (int32_t)all;   // wrapping silently)
        if (total_copy) total_copy[0] = total_out[0];   // (the ctx's count word of an asynchronous step: no copy launch)
        // and straight into the ctx's pinned host word (round 6: the 4-byte hipMemcpyAsync that used to carry it cost a blit
        // launch and ~20 us of drained GPU around it, every step); the host reads it behind the event recorded after this
        // kernel, never earlier
        if (total_host) __hip_atomic_store(total_host, total_out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int tile_excl = (int)(uint32_t)excl;
    int run = tile_excl + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        run += v[i];
        tile[SCAN_PAD(threadIdx.x * SCAN_ITEMS + i)] = run;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int e = i * SCAN_THREADS + threadIdx.x;
        const int64_t idx = base0 + e;
        if (idx < n) {
            out[idx] = tile[SCAN_PAD(e)];
            if (PACK) {
                const uint32_t ex = (uint32_t)(e == 0 ? tile_excl : tile[SCAN_PAD(e - 1)]);
                pack_out[idx] = ((uint64_t)ex << 32) | geo[i];
            }
        }
    }
}

// Control block of the single-pass scan in one arena slot: two ticket counters, four rotating totals, then one status
// word per tile.  Nothing is cleared between launches (generation-stamped words, the ticket counters take turns); the
// host clears the block when it is (re)allocated and when the 14-bit generation is about to repeat.
struct ChainCtl { uint32_t* tickets; int32_t* total; u64* status; uint32_t gen; };
static int chain_ctl(st3r_ctx* ctx, hipStream_t s, int64_t nwords, ChainCtl* c) {
    void* p; int grown = 0;
    int rc = st3r_arena_get2(ctx, SLOT_SCAN_CHAIN, sizeof(u64) * ((size_t)nwords + 16), &p, &grown);
    if (rc) return rc;
    if (grown || ctx->scan_gen >= 0x3FFF) {
        HIP_TRY(hipMemsetAsync(p, 0, ctx->slot_bytes[SLOT_SCAN_CHAIN], s));
        ctx->scan_gen = 0;
    }
    u64* ctl = (u64*)p;
    c->gen = ++ctx->scan_gen;
    c->tickets = (uint32_t*)ctl;                           // ctl[0..3]
    c->total = (int32_t*)(ctl + 4) + (c->gen & 3);         // ctl[4..5]
    c->status = ctl + 16;
    return ST3R_OK;
}

// out = inclusive scan of in (counts) or of the areas of the packed rectangles `rects`; the grand total is left in
// device memory (*total_dev)
int st3r_scan_inclusive_i32(st3r_ctx* ctx, hipStream_t s, const int32_t* in, const void* rects, int rect32, int32_t* out,
                            int64_t n, int32_t** total_dev, uint64_t* pack_out = nullptr, int32_t* total_copy = nullptr,
                            int32_t* total_host = nullptr) {
    const int ntiles = ceil_div(n, SCAN_TILE);
    ChainCtl c;
    int rc = chain_ctl(ctx, s, ntiles, &c);
    if (rc) return rc;
#define SCAN_LAUNCH(R, P)                                                                                                \
    hipLaunchKernelGGL((k_scan_chained<R, P>), dim3(ntiles), dim3(SCAN_THREADS), 0, s, in, rects, rect32, n, out, pack_out, \
                       c.status, c.tickets, c.gen, c.total, total_copy, total_host)
    if (rects && pack_out) SCAN_LAUNCH(true, true);
    else if (rects) SCAN_LAUNCH(true, false);
    else SCAN_LAUNCH(false, false);
#undef SCAN_LAUNCH
    LAUNCH_CHECK();
    if (total_dev) *total_dev = c.total;
    return ST3R_OK;
}

// Pair-id order scan.  tiles: the counts (stage API), or NULL with pack_rects = the pairs' packed rectangles (fused
// path; pack_out then also receives slot base | rectangle per pair when it is not NULL).
int st3r_isect_scan_impl(st3r_ctx* ctx, hipStream_t s, int64_t n_pairs, const int32_t* tiles, int32_t* cum,
                         int64_t* n_isects_host, const void* pack_rects, int rect32, uint64_t* pack_out,
                         int32_t** total_dev_out, int32_t* total_copy, int32_t* total_host) {
    if (n_pairs == 0) { if (n_isects_host) *n_isects_host = 0; return ST3R_OK; }
    int32_t* total_dev = nullptr;
    int rc = st3r_scan_inclusive_i32(ctx, s, tiles, tiles ? nullptr : pack_rects, rect32, cum, n_pairs, &total_dev,
                                     tiles ? nullptr : pack_out, total_copy, total_host);
    if (rc) return rc;
    if (total_dev_out) *total_dev_out = total_dev;
    if (n_isects_host) {
        int32_t* pin = (int32_t*)ctx->pinned;
        HIP_TRY(hipMemcpyAsync(pin, total_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *n_isects_host = (int64_t)pin[0];
    }
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_isect_scan(st3r_ctx* ctx, void* stream, int64_t n_pairs, const int32_t* tiles_per_gauss,
                                   int32_t* cum_tiles, int64_t* n_isects_host) {
    ARG_CHECK(ctx && n_pairs >= 0 && tiles_per_gauss && cum_tiles && n_isects_host);
    return st3r_isect_scan_impl(ctx, (hipStream_t)stream, n_pairs, tiles_per_gauss, cum_tiles, n_isects_host, nullptr,
                                0, nullptr, nullptr, nullptr, nullptr);
}

__global__ __launch_bounds__(256) void k_isect_emit(int N, int64_t n_pairs, const float4* __restrict__ splats,
                                                    const int32_t* __restrict__ cum, int tile_size, int tile_w,
                                                    int tile_h, int tile_n_bits, int64_t* __restrict__ isect_ids,
                                                    int32_t* __restrict__ flatten_ids) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n_pairs) return;
    const int end = cum[pid];
    const int start = pid == 0 ? 0 : cum[pid - 1];
    if (end == start) return;
    const float4 r0 = splats[pid * 3 + 0];
    const float4 r2 = splats[pid * 3 + 2];
    const float radius = (float)__float_as_int(r2.z);
    const TileRect tr = ref_tile_rect(r0.x, r0.y, radius, tile_size, tile_w, tile_h);
    const int x0 = tr.x0, y0 = tr.y0, x1 = tr.x1, y1 = tr.y1;
    const int64_t cid = pid / N;
    const int64_t cid_enc = cid << (32 + tile_n_bits);
    const int64_t depth_enc = (int64_t)(uint32_t)__float_as_int(r2.y);
    int cur = start;
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            const int64_t tile_id = (int64_t)ty * tile_w + tx;
            isect_ids[cur] = cid_enc | (tile_id << 32) | depth_enc;
            flatten_ids[cur] = (int32_t)pid;
            ++cur;
        }
}

static int bit_length_u32(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

int st3r_isect_emit_impl(hipStream_t s, int N, int C, const float* splats, const int32_t* cum, int tile_size,
                         int tile_w, int tile_h, int64_t* isect_ids, int32_t* flatten_ids) {
    int64_t n_pairs = (int64_t)N * C;
    if (n_pairs == 0) return ST3R_OK;
    int tile_n_bits = bit_length_u32((uint32_t)(tile_w * tile_h));
    hipLaunchKernelGGL(k_isect_emit, dim3(ceil_div(n_pairs, 256)), dim3(256), 0, s, N, n_pairs, (const float4*)splats,
                       cum, tile_size, tile_w, tile_h, tile_n_bits, isect_ids, flatten_ids);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_isect_emit(st3r_ctx* ctx, void* stream, int N, int C, const float* splats,
                                   const int32_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                                   int64_t n_isects, int64_t* isect_ids, int32_t* flatten_ids) {
    ARG_CHECK(ctx && N >= 0 && C > 0 && splats && cum_tiles && tile_size > 0 && tile_w > 0 && tile_h > 0);
    if (n_isects == 0) return ST3R_OK;
    ARG_CHECK(isect_ids && flatten_ids);
    return st3r_isect_emit_impl((hipStream_t)stream, N, C, splats, cum_tiles, tile_size, tile_w, tile_h, isect_ids,
                                flatten_ids);
}

__global__ __launch_bounds__(256) void k_isect_offsets(int64_t n_isects, const int64_t* __restrict__ ids, int C,
                                                       int n_tiles, int tile_n_bits,
                                                       int32_t* __restrict__ offsets) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_isects) return;
    const int64_t hi = ids[idx] >> 32;
    const int64_t id_curr = (hi >> tile_n_bits) * n_tiles + (hi & (((int64_t)1 << tile_n_bits) - 1));
    if (idx == 0) {
        for (int64_t i = 0; i <= id_curr; ++i) offsets[i] = 0;
    } else {
        const int64_t hp = ids[idx - 1] >> 32;
        const int64_t id_prev = (hp >> tile_n_bits) * n_tiles + (hp & (((int64_t)1 << tile_n_bits) - 1));
        if (id_prev != id_curr)
            for (int64_t i = id_prev + 1; i <= id_curr; ++i) offsets[i] = (int32_t)idx;
    }
    if (idx == n_isects - 1) {
        const int64_t total = (int64_t)C * n_tiles;
        for (int64_t i = id_curr + 1; i < total; ++i) offsets[i] = (int32_t)n_isects;
    }
}

int st3r_isect_offsets_impl(hipStream_t s, int64_t n_isects, const int64_t* ids, int C, int tile_w, int tile_h,
                            int32_t* offsets) {
    int n_tiles = tile_w * tile_h;
    if (n_isects == 0) {
        HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)C * n_tiles, s));
        return ST3R_OK;
    }
    int tile_n_bits = bit_length_u32((uint32_t)n_tiles);
    hipLaunchKernelGGL(k_isect_offsets, dim3(ceil_div(n_isects, 256)), dim3(256), 0, s, n_isects, ids, C, n_tiles,
                       tile_n_bits, offsets);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_offsets(st3r_ctx* ctx, void* stream, int64_t n_isects, const int64_t* isect_ids_sorted,
                                int C, int tile_w, int tile_h, int32_t* offsets) {
    ARG_CHECK(ctx && n_isects >= 0 && C > 0 && tile_w > 0 && tile_h > 0 && offsets);
    ARG_CHECK(n_isects == 0 || isect_ids_sorted);
    return st3r_isect_offsets_impl((hipStream_t)stream, n_isects, isect_ids_sorted, C, tile_w, tile_h, offsets);
}

// ------------------------------------------------------------------------------------
// Two-level sort of the fused path (same final order as the 64-bit key sort above):
//   1. pairs are sorted by (camera | depth bits) once            -> perm[s] = pair id
//   2. records are emitted in that order with a 32-bit key        camera * n_tiles + tile
//   3. a STABLE sort on that key groups them by (camera, tile) and keeps depth order inside
// Ties (same camera, tile, depth bits) keep pair-id order in both schemes: step 1 is stable
// on pair id, steps 2/3 preserve it.
// ------------------------------------------------------------------------------------
// offsets[k] = first sorted position whose 32-bit (camera, tile) key is >= k
__global__ __launch_bounds__(256) void k_isect_offsets32(int64_t n_isects, const uint32_t* __restrict__ keys,
                                                         int64_t total, int32_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ n_dev) {
    // n_dev: the record count lives on the device (n_isects is then the launch capacity); offsets has total + 1
    // entries, the last one = the record count (end of the last tile for the blend kernels).  Four keys per thread
    // (one 16-byte load) plus the key in front of them.
    if (n_dev) n_isects = min(n_isects, (int64_t)max(*n_dev, 0));
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_isects == 0) {
        for (int64_t i = tid; i <= total; i += (int64_t)gridDim.x * blockDim.x) offsets[i] = 0;
        return;
    }
    const int64_t i0 = tid * 4;
    if (i0 >= n_isects) return;
    uint32_t k[4];
    if (i0 + 4 <= n_isects) {
        const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
        k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = i0 + j < n_isects ? keys[i0 + j] : 0u;
    }
    int64_t prev = i0 == 0 ? -1 : (int64_t)keys[i0 - 1];   // tiles before the first key start at 0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t idx = i0 + j;
        if (idx < n_isects) {
            const int64_t cur = k[j];
            for (int64_t i = prev + 1; i <= cur; ++i) offsets[i] = (int32_t)idx;   // (empty when cur == prev)
            prev = cur;
            if (idx == n_isects - 1)
                for (int64_t i = cur + 1; i <= total; ++i) offsets[i] = (int32_t)n_isects;
        }
    }
}

// offsets: [C*tiles + 1] (one more than gsplat's table: the total closes the last tile)
int st3r_isect_offsets32_impl(hipStream_t s, int64_t n_isects, const uint32_t* keys, int C, int tile_w, int tile_h,
                              int32_t* offsets, const int32_t* n_dev) {
    const int64_t total = (int64_t)C * tile_w * tile_h;
    if (n_isects == 0 && !n_dev) {
        HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)(total + 1), s));
        return ST3R_OK;
    }
    hipLaunchKernelGGL(k_isect_offsets32, dim3(ceil_div(n_isects > 0 ? (n_isects + 3) / 4 : 1, 256)), dim3(256), 0, s, n_isects,
                       keys, total, offsets, n_dev);
    LAUNCH_CHECK();
    return ST3R_OK;
}

// ------------------------------------------------------------------------------------
// Rasterizing from records that another rank projected (Gaussian-sharded multi-GPU mode): rebuild what
// k_project_sh_fwd would have written next to them -- the tile count (same rectangle code) and the
// (camera | depth bits) key of the two-level sort.  A record with radius 0 is a culled pair.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_records_prepare(int N, int64_t n_pairs, const float4* __restrict__ splats,
                                                         int tile_size, int tile_w, int tile_h, int tight,
                                                         int32_t* __restrict__ tiles, uint64_t* __restrict__ depth_keys,
                                                         int32_t* __restrict__ depth_vals, uint32_t key_base,
                                                         void* __restrict__ rects, int rect32) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n_pairs) return;
    const float4 r2 = splats[pid * 3 + 2];
    const int radius = __float_as_int(r2.z);
    int ntiles = 0;
    TileRect tr = {0, 0, 0, 0};
    if (radius > 0) {
        const float4 r0 = splats[pid * 3 + 0];
        tr = ref_tile_rect(r0.x, r0.y, (float)radius, tile_size, tile_w, tile_h);
        if (tight) {
            const float4 r1 = splats[pid * 3 + 1];
            tr = tight_tile_rect(tr, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
        }
        ntiles = (tr.y1 - tr.y0) * (tr.x1 - tr.x0);
    }
    if (tiles) tiles[pid] = ntiles;
    if (rects) rect_store(rects, rect32, pid, tr);
    const uint32_t dbits = radius > 0 ? (uint32_t)__float_as_int(r2.y) : 0xFFFFFFFFu;
    if (key_base)   // same packed key as k_project_sh_fwd
        reinterpret_cast<uint32_t*>(depth_keys)[pid] =
            ((uint32_t)(pid / N) << 29) | (radius > 0 ? dbits - key_base : 0x1FFFFFFFu);
    else
        depth_keys[pid] = ((uint64_t)(pid / N) << 32) | dbits;
    depth_vals[pid] = (int32_t)pid;
}

int st3r_records_prepare_impl(hipStream_t s, int N, int C, const float* splats, int tile_size, int tile_w, int tile_h,
                              int tight, int32_t* tiles, uint64_t* depth_keys, int32_t* depth_vals, uint32_t key_base,
                              void* rects, int rect32) {
    const int64_t n_pairs = (int64_t)N * C;
    if (n_pairs == 0) return ST3R_OK;
    hipLaunchKernelGGL(k_records_prepare, dim3(ceil_div(n_pairs, 256)), dim3(256), 0, s, N, n_pairs,
                       (const float4*)splats, tile_size, tile_w, tile_h, tight, tiles, depth_keys, depth_vals, key_base,
                       rects, rect32);
    LAUNCH_CHECK();
    return ST3R_OK;
}


// Emission of the fused path: records leave in (camera, depth) order, so that the stable (camera, tile) sort behind
// these kernels keeps depth order inside every tile.  Two launches (round 3: four -- a depth-order scan made of gather +
// reduce, tile sums and downsweep, then the emission; rounds 4-5: three):
//   k_isect_gather   a workgroup owns EP consecutive pairs of ONE camera in depth order (perm[] from the level-1 sort)
//                    and gathers their packed rectangles by pair id -- the only random access of the front end -- into
//                    depth order, plus the workgroup's tile count.  With eight views (or a multiple) the workgroups of
//                    a camera run on one XCD (workgroup b runs on XCD b % 8), whose L2 then serves that camera's
//                    rectangles: 4 MB at 1 M Gaussians in the 32-bit form;
//   k_isect_emit_d   the same workgroup shape: scans its pairs' tile counts (rectangle areas), finds its first output
//                    position as the tile counts of all earlier workgroups, camera-major (summed by the workgroup itself;
//                    k_isect_wg_scan, a launch between the two, only for very large grids), and emits.  The record count
//                    is the pair-order scan's total.  Work is dealt by OUTPUT element, not by pair:
//                    every pair with tiles marks its first output with its index, a prefix maximum over the chunk
//                    spreads the marks to the right (thread t scans 16 consecutive entries in registers, the thread
//                    maxima meet in one workgroup scan), then the outputs are dealt to the threads with a stride of 256
//                    and decode their tile from the index inside the owner's rectangle -- coalesced stores, no
//                    per-thread loops over rectangles of very different sizes.
// Measured on the way (SYNTH-1M): ONE kernel doing gather + chained scan with decoupled look-back per camera + emission
// took 0.22 ms, of which 0.09 ms were workgroups waiting for the slowest gather among their predecessors (0.13 ms with
// the look-back compiled out, 0.15 ms with sequential instead of gathered rectangles); a binary search over the scan
// entries per output (round 3's emission) costs ten dependent LDS reads.
// Everything the emission reads is sequential; no floating point except one reciprocal for the row of a tile.
// cap: capacity of tile_keys / vals.  With the record count on the device the buffers are sized from the previous
// step's count; records past the capacity are dropped here (the host notices the overflow when it reads the count
// back before the next step and fails loudly) -- never written out of bounds.
#ifndef EP
#define EP 1024               // pairs per workgroup
#endif
#define EPT (EP / 256)        // pairs per thread
#define ECH 4096              // outputs per emission chunk (16 per thread)

// workgroup -> (camera, chunk of EP pairs).  Views a multiple of 8: workgroup b serves a camera c = b (mod 8).
__device__ __forceinline__ void emit_item(int C, int bpc, int* c, int* k) {
    const int b = blockIdx.x;
    if ((C & 7) == 0) { const int j = b >> 3; *c = (b & 7) + 8 * (j / bpc); *k = j % bpc; }
    else { *c = b / bpc; *k = b % bpc; }
}

__global__ __launch_bounds__(256) void k_isect_gather(int N, int C, int bpc, const int32_t* __restrict__ perm,
                                                      const void* __restrict__ rects, int rect32,
                                                      void* __restrict__ rects_d, int32_t* __restrict__ wg_tiles) {
    int c, k;
    emit_item(C, bpc, &c, &k);
    const int t = threadIdx.x;
    const int64_t first = (int64_t)c * N + (int64_t)k * EP;
    const int np = (int)min((int64_t)EP, (int64_t)(c + 1) * N - first);
    int32_t pid[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) pid[j] = j * 256 + t < np ? perm[first + j * 256 + t] : -1;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        if (pid[j] >= 0) {
            const int64_t dst = first + j * 256 + t;
            if (rect32) {
                const uint32_t r = reinterpret_cast<const uint32_t*>(rects)[pid[j]];
                reinterpret_cast<uint32_t*>(rects_d)[dst] = r;
                sum += (int)((r >> 16) & 0xFFu) * (int)(r >> 24);
            } else {
                const uint64_t r = reinterpret_cast<const uint64_t*>(rects)[pid[j]];
                reinterpret_cast<uint64_t*>(rects_d)[dst] = r;
                sum += (int)((r >> 32) & 0xFFFF) * (int)(r >> 48);
            }
        }
    }
    int total;
    block_incl_scan(sum, &total);
    if (t == 0) wg_tiles[c * bpc + k] = total;   // < 2^30: 1024 rectangles of < 2^20 tiles
}

// Exclusive prefix of the workgroup tile counts over ALL cameras, camera-major, as a launch of its own: only for grids of
// more than EMIT_SELF_BASE_MAX workgroups (5 M Gaussians x 8 views: 39 k).  Below that every emission workgroup adds up
// the counts of its predecessors itself (k_isect_emit_d with wg_base = NULL: at most 64 KB of L2 reads per workgroup) and
// this launch -- one workgroup, 13 us with the whole GPU drained around it -- does not exist (round 6).
#define EMIT_SELF_BASE_MAX 16384
__global__ __launch_bounds__(1024) void k_isect_wg_scan(int n, const int32_t* __restrict__ wg_tiles,
                                                        long long* __restrict__ wg_base) {
    // thread t owns the consecutive entries [t per, (t + 1) per): their sum, one scan over the 1024 sums, then the
    // entries again (a loop of block scans over 256 entries at a time took 45 us here: 31 dependent rounds)
    __shared__ long long s_wave[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (n + 1023) / 1024;
    const int e0 = min(t * per, n), e1 = min(e0 + per, n);
    long long sum = 0;
    for (int e = e0; e < e1; ++e) sum += wg_tiles[e];
    long long inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    long long base = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) base += i < w ? s_wave[i] : 0;
    long long run = base + inc - sum;
    for (int e = e0; e < e1; ++e) { wg_base[e] = run; run += wg_tiles[e]; }
}

__global__ __launch_bounds__(256) void k_isect_emit_d(int N, int C, int bpc, const int32_t* __restrict__ perm,
                                                      const void* __restrict__ rects_d, int rect32,
                                                      const long long* __restrict__ wg_base,
                                                      const int32_t* __restrict__ wg_tiles, int tile_w, int tile_h,
                                                      uint32_t* __restrict__ tile_keys, int32_t* __restrict__ vals,
                                                      int64_t cap) {
    __shared__ int s_end[EP];         // inclusive scan of the workgroup's tile counts
    __shared__ uint32_t s_org[EP];    // x0 | y0 << 16
    __shared__ uint16_t s_w[EP];      // rectangle width (< 2^16 in both packed forms; 22.6 KB of LDS in all: seven workgroups per CU)
    __shared__ int32_t s_pid[EP];
    __shared__ uint16_t s_own[ECH];   // owner (pair index + 1) of every output of the current emission chunk
    __shared__ unsigned s_wmax[4];
    __shared__ long long s_bsum[4];
    int c, k;
    emit_item(C, bpc, &c, &k);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // first output position: the tile counts of all earlier workgroups, camera-major.  Summed here (the loads travel
    // under the rectangle loads and the scan below) unless the grid is large enough for a scan launch to pay.
    long long bsum = 0;
    if (!wg_base) {
        const int nb4 = (c * bpc + k) >> 2;   // whole 16-byte groups in front of this workgroup's entry
        for (int i = t; i < nb4; i += 256) {
            const int4 v = reinterpret_cast<const int4*>(wg_tiles)[i];
            bsum += ((long long)v.x + v.y) + ((long long)v.z + v.w);
        }
        if (t < ((c * bpc + k) & 3)) bsum += wg_tiles[4 * nb4 + t];
    }
    const int64_t first = (int64_t)c * N + (int64_t)k * EP;
    const int np = (int)min((int64_t)EP, (int64_t)(c + 1) * N - first);
    int cnt[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = j * 256 + t;
        int32_t pid = 0;
        uint32_t org = 0, rw = 0, rh = 0;
        if (e < np) {
            pid = perm[first + e];
            rect_load(rects_d, rect32, first + e, &org, &rw, &rh);
        }
        s_pid[e] = pid; s_org[e] = org; s_w[e] = (uint16_t)rw;
        cnt[j] = (int)(rw * rh);
    }
    // scan in pair order: EPT block scans of 256 consecutive pairs, the carry in a register
    int carry = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        int tot;
        const int inc = block_incl_scan(cnt[j], &tot);
        s_end[j * 256 + t] = carry + inc;
        carry += tot;
    }
    const int total = carry;
    __syncthreads();
    // (a true count above 2^31 wraps the int32 pair-order scan: the position test below keeps such writes out)
    long long base;
    if (wg_base) base = wg_base[c * bpc + k];
    else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bsum += __shfl_down(bsum, off);
        if (lane == 0) s_bsum[w] = bsum;
        __syncthreads();
        base = (s_bsum[0] + s_bsum[1]) + (s_bsum[2] + s_bsum[3]);
    }
    const uint32_t key0 = (uint32_t)c * (uint32_t)(tile_w * tile_h);
    int carry_owner = 0;   // owner (+1) of the last output of the previous chunk (uniform)
    for (int c0 = 0; c0 < total; c0 += ECH) {
        reinterpret_cast<uint4*>(s_own)[t] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(s_own)[t + 256] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int e = j * 256 + t;
            if (e < np) {
                const int start = e ? s_end[e - 1] : 0;
                if (s_end[e] > start && start >= c0 && start < c0 + ECH) s_own[start - c0] = (uint16_t)(e + 1);
            }
        }
        __syncthreads();
        uint4 a = reinterpret_cast<const uint4*>(s_own)[2 * t], b = reinterpret_cast<const uint4*>(s_own)[2 * t + 1];
        unsigned wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned run = 0;   // running maximum over this thread's 16 entries
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned lo = max(run, wv[i] & 0xFFFFu);
            run = max(lo, wv[i] >> 16);
            wv[i] = lo | (run << 16);
        }
        unsigned inc = run;   // inclusive prefix maximum over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off);
            if (lane >= off) inc = max(inc, o);
        }
        if (lane == 63) s_wmax[w] = inc;
        unsigned excl_m = __shfl_up(inc, 1);
        if (lane == 0) excl_m = 0;
        __syncthreads();
        unsigned pre = max((unsigned)carry_owner, excl_m);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww)
            if (ww < w) pre = max(pre, s_wmax[ww]);
        carry_owner = (int)max(max(max((unsigned)carry_owner, s_wmax[0]), max(s_wmax[1], s_wmax[2])), s_wmax[3]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned lo = max(wv[i] & 0xFFFFu, pre), hi = max(wv[i] >> 16, pre);
            wv[i] = lo | (hi << 16);
        }
        reinterpret_cast<uint4*>(s_own)[2 * t] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        reinterpret_cast<uint4*>(s_own)[2 * t + 1] = make_uint4(wv[4], wv[5], wv[6], wv[7]);
        __syncthreads();
        const int lim = min(ECH, total - c0);
        for (int i = t; i < lim; i += 256) {
            const int p = (int)s_own[i] - 1;
            const int o = c0 + i;
            const int kk = o - (p == 0 ? 0 : s_end[p - 1]);   // index among the pair's emitted tiles, row major
            const uint32_t w_ = s_w[p], org = s_org[p];
            // kk / w with kk < w * h <= 2^20 and w < 2^10: float estimate, corrected by one either way
            int qq = (int)((float)kk * __builtin_amdgcn_rcpf((float)w_));
            int rem = kk - qq * (int)w_;
            if (rem < 0) { --qq; rem += (int)w_; }
            if (rem >= (int)w_) { ++qq; rem -= (int)w_; }
            const uint32_t tx = (org & 0xFFFF) + (uint32_t)rem, ty = (org >> 16) + (uint32_t)qq;
            const long long pos = base + o;
            if (pos >= 0 && pos < cap) {
                tile_keys[pos] = key0 + ty * (uint32_t)tile_w + tx;
                vals[pos] = s_pid[p];
            }
        }
        __syncthreads();
    }
}

int st3r_isect_emit_chain_impl(st3r_ctx* ctx, hipStream_t s, int N, int C, const int32_t* perm, const void* rects,
                               int rect32, int tile_w, int tile_h, uint32_t* tile_keys, int32_t* vals, int64_t cap) {
    const int64_t n_pairs = (int64_t)N * C;
    if (n_pairs == 0) return ST3R_OK;
    const int bpc = ceil_div(N, EP);
    const int grid = C * bpc;
    void* p;
    int rc = st3r_arena_get(ctx, SLOT_RECTS_D, (rect32 ? sizeof(uint32_t) : sizeof(uint64_t)) * (size_t)n_pairs, &p);
    if (rc) return rc;
    void* rects_d = p;
    rc = st3r_arena_get(ctx, SLOT_CUM_D, (sizeof(long long) + sizeof(int32_t)) * (size_t)grid + 16, &p);
    if (rc) return rc;
    long long* wg_base = (long long*)p;
    int32_t* wg_tiles = (int32_t*)(wg_base + grid);   // (8-byte entries in front: 16-byte aligned for even grids -- see below)
    if (((uintptr_t)wg_tiles & 15) != 0) wg_tiles += 2;
    hipLaunchKernelGGL(k_isect_gather, dim3(grid), dim3(256), 0, s, N, C, bpc, perm, rects, rect32, rects_d, wg_tiles);
    const bool self_base = grid <= EMIT_SELF_BASE_MAX;
    if (!self_base) hipLaunchKernelGGL(k_isect_wg_scan, dim3(1), dim3(1024), 0, s, grid, wg_tiles, wg_base);
    hipLaunchKernelGGL(k_isect_emit_d, dim3(grid), dim3(256), 0, s, N, C, bpc, perm, rects_d, rect32,
                       self_base ? (const long long*)nullptr : wg_base, wg_tiles, tile_w, tile_h, tile_keys, vals, cap);
    LAUNCH_CHECK();
    return ST3R_OK;
}
