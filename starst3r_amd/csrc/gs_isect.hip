// K2/K3/K5: tile-intersection bookkeeping.
//   scan    -- inclusive prefix sum of tiles_per_gauss (gsplat: torch.cumsum between the two
//              isect_tiles passes), three-phase: per-block reduce, one-block scan of the
//              block sums, per-block downsweep with wave64 prefix scans;
//   emit    -- gsplat isect_tiles pass 2: one (key,value) per touched tile;
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

// `perm` (optional): scan in[perm[idx]] instead of in[idx] (tile counts visited in depth order)
// `rects` (optional, with perm): packed tile rectangles (x0 | y0 << 16 | w << 32 | h << 48) instead of counts; the
// gathered rectangles are stored in depth order (rects_sorted) for the emit kernel, which then streams.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const int32_t* __restrict__ in,
                                                              const int32_t* __restrict__ perm, int64_t n,
                                                              int32_t* __restrict__ block_sums,
                                                              int32_t* __restrict__ gathered,
                                                              const uint64_t* __restrict__ rects,
                                                              uint64_t* __restrict__ rects_sorted) {
    // Gathering launch: workgroup b runs on XCD b % 8 (each XCD has its own L2).  XCD x takes the x-th eighth of the
    // array, i.e. (with 8 views) the pairs of one camera in depth order: its random gathers stay inside that camera's
    // 8 MB of rectangles instead of every L2 streaming all of them.
    int bid = blockIdx.x;
    if (perm) {   // measured at SYNTH-1M: 167 -> 107 us
        const int G = gridDim.x >> 3;
        if (bid < (G << 3)) bid = (bid & 7) * G + (bid >> 3);
    }
    const int64_t base = (int64_t)bid * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        int64_t idx = base + (int64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) {
            int v;
            if (rects) {
                const uint64_t r = rects[perm[idx]];
                rects_sorted[idx] = r;
                v = (int)((r >> 32) & 0xFFFF) * (int)(r >> 48);
                gathered[idx] = v;
            } else if (perm) { v = in[perm[idx]]; gathered[idx] = v; }  // the downsweep then reads contiguously
            else v = in[idx];
            s += v;
        }
    }
    int total;
    block_incl_scan(s, &total);
    if (threadIdx.x == 0) block_sums[bid] = total;
}

// single block: exclusive scan of block_sums in place; grand total -> total_out[0]
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocksums(int32_t* __restrict__ block_sums, int nblocks,
                                                                 int32_t* __restrict__ total_out) {
    // the running total is kept in 64 bits: a grand total past 2^31 - 1 is reported as -1 (the callers turn a
    // negative count into ST3R_ERR_INVALID) instead of wrapping silently
    int64_t carry = 0;
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int idx = base + threadIdx.x;
        int v = idx < nblocks ? block_sums[idx] : 0;
        int total;
        int inc = block_incl_scan(v, &total);
        if (idx < nblocks) block_sums[idx] = (int)carry + inc - v;
        carry += (int64_t)(uint32_t)total;   // a block total < 2^32 (4096 counts < 2^20 each)
    }
    if (threadIdx.x == 0) total_out[0] = carry > 2147483647LL ? -1 : (int32_t)carry;
}

// (in may alias out: the block loads its whole tile before it stores any of it)
// Thread t owns SCAN_ITEMS consecutive elements so the scan order is the array order; the tile travels through LDS so
// that both the loads and the stores are coalesced (padded by one word per 32: the 16-word runs of neighbouring
// threads would otherwise meet in the same banks).
#define SCAN_PAD(i) ((i) + ((i) >> 5))
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_down(const int32_t* in, const int32_t* __restrict__ perm,
                                                            int64_t n, const int32_t* __restrict__ block_sums,
                                                            int32_t* out, const uint64_t* __restrict__ pack_rects,
                                                            uint64_t* __restrict__ pack_out) {
    // pack_rects / pack_out (optional): also leave, per element, exclusive prefix << 32 | x0 | y0 << 10 | w << 20 of its
    // packed tile rectangle -- the one word the blend backward gathers per staged record (slot base + rectangle; two
    // separate random reads cost a 64-byte sector each)
    __shared__ int tile[SCAN_PAD(SCAN_TILE) + 1];
    const int64_t base0 = (int64_t)blockIdx.x * SCAN_TILE;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int e = i * SCAN_THREADS + threadIdx.x;
        const int64_t idx = base0 + e;
        tile[SCAN_PAD(e)] = idx < n ? (perm ? in[perm[idx]] : in[idx]) : 0;
    }
    __syncthreads();
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int e = threadIdx.x * SCAN_ITEMS + i;
        v[i] = tile[SCAN_PAD(e)];
        s += v[i];
    }
    int total;
    int inc = block_incl_scan(s, &total);
    int run = block_sums[blockIdx.x] + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        run += v[i];
        tile[SCAN_PAD(threadIdx.x * SCAN_ITEMS + i)] = run;   // own elements only: no barrier needed before
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int e = i * SCAN_THREADS + threadIdx.x;
        const int64_t idx = base0 + e;
        if (idx < n) {
            out[idx] = tile[SCAN_PAD(e)];
            if (pack_out) {
                const uint32_t excl = (uint32_t)(e == 0 ? block_sums[blockIdx.x] : tile[SCAN_PAD(e - 1)]);
                const uint64_t r = pack_rects[idx];
                const uint32_t geo = (uint32_t)(r & 0x3FF) | ((uint32_t)((r >> 16) & 0x3FF) << 10) |
                                     ((uint32_t)((r >> 32) & 0x3FF) << 20);
                pack_out[idx] = ((uint64_t)excl << 32) | geo;
            }
        }
    }
}

// out = inclusive scan(in); the grand total is left in the SLOT_SCAN_TMP buffer at [nblocks]
int st3r_scan_inclusive_i32(st3r_ctx* ctx, hipStream_t s, const int32_t* in, const int32_t* perm, int32_t* out,
                            int64_t n, int32_t** total_dev, const uint64_t* rects = nullptr,
                            uint64_t* rects_sorted = nullptr, const uint64_t* pack_rects = nullptr,
                            uint64_t* pack_out = nullptr) {
    int nblocks = ceil_div(n, SCAN_TILE);
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SCAN_TMP, sizeof(int32_t) * (size_t)(nblocks + 4), &tmp);
    if (rc) return rc;
    int32_t* bs = (int32_t*)tmp;
    // with a permutation the gathered values are parked in `out` by the reduce pass and scanned in place
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblocks), dim3(SCAN_THREADS), 0, s, in, perm, n, bs, out, rects, rects_sorted);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_THREADS), 0, s, bs, nblocks, bs + nblocks);
    hipLaunchKernelGGL(k_scan_down, dim3(nblocks), dim3(SCAN_THREADS), 0, s, perm ? out : in,
                       (const int32_t*)nullptr, n, bs, out, pack_rects, pack_out);
    LAUNCH_CHECK();
    if (total_dev) *total_dev = bs + nblocks;
    return ST3R_OK;
}

// inclusive scan of tiles[perm[.]] (perm may be NULL); optional synchronous read-back of the total
int st3r_isect_scan_perm_impl(st3r_ctx* ctx, hipStream_t s, int64_t n_pairs, const int32_t* tiles,
                              const int32_t* perm, int32_t* cum, int32_t** total_dev_out, const uint64_t* rects,
                              uint64_t* rects_sorted) {
    if (n_pairs == 0) return ST3R_OK;
    return st3r_scan_inclusive_i32(ctx, s, tiles, perm, cum, n_pairs, total_dev_out, rects, rects_sorted);
}

int st3r_isect_scan_impl(st3r_ctx* ctx, hipStream_t s, int64_t n_pairs, const int32_t* tiles, int32_t* cum,
                         int64_t* n_isects_host, const uint64_t* pack_rects, uint64_t* pack_out) {
    if (n_pairs == 0) { if (n_isects_host) *n_isects_host = 0; return ST3R_OK; }
    int32_t* total_dev = nullptr;
    int rc = st3r_scan_inclusive_i32(ctx, s, tiles, nullptr, cum, n_pairs, &total_dev, nullptr, nullptr, pack_rects,
                                     pack_out);
    if (rc) return rc;
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
                                nullptr);
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
                                                         uint64_t* __restrict__ rects) {
    const int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n_pairs) return;
    const float4 r2 = splats[pid * 3 + 2];
    const int radius = __float_as_int(r2.z);
    int ntiles = 0;
    uint64_t rect = 0;
    if (radius > 0) {
        const float4 r0 = splats[pid * 3 + 0];
        TileRect tr = ref_tile_rect(r0.x, r0.y, (float)radius, tile_size, tile_w, tile_h);
        if (tight) {
            const float4 r1 = splats[pid * 3 + 1];
            tr = tight_tile_rect(tr, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
        }
        ntiles = (tr.y1 - tr.y0) * (tr.x1 - tr.x0);
        rect = pack_rect(tr);
    }
    tiles[pid] = ntiles;
    if (rects) rects[pid] = rect;
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
                              uint64_t* rects) {
    const int64_t n_pairs = (int64_t)N * C;
    if (n_pairs == 0) return ST3R_OK;
    hipLaunchKernelGGL(k_records_prepare, dim3(ceil_div(n_pairs, 256)), dim3(256), 0, s, N, n_pairs,
                       (const float4*)splats, tile_size, tile_w, tile_h, tight, tiles, depth_keys, depth_vals, key_base,
                       rects);
    LAUNCH_CHECK();
    return ST3R_OK;
}

// Emission from the depth-ordered packed rectangles (written by the depth-order scan): everything this kernel
// reads is sequential -- no gather of 48-byte records, no floating point.  The 256 pairs of a block own one
// contiguous output range.  Work is dealt by OUTPUT element, not by pair: a thread finds the pair that owns its
// element with a binary search over the block's 256 scan entries (LDS) and decodes the tile from the element's index
// inside the pair's rectangle -- coalesced stores, no per-thread loops over rectangles of very different sizes
// (one thread per pair with the range assembled in LDS: 0.123 ms at SYNTH-1M, this: 0.083 ms), any range length.
__global__ __launch_bounds__(256) void k_isect_emit_rects(int N, int64_t n_pairs, const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ cum_sorted,
                                                          const uint64_t* __restrict__ rects_sorted, int tile_w,
                                                          int tile_h, uint32_t* __restrict__ tile_keys,
                                                          int32_t* __restrict__ vals, int64_t cap) {
    // cap: capacity of tile_keys / vals.  With the record count on the device the buffers are sized from the previous
    // step's count; records past the capacity are dropped here (the host notices the overflow when it reads the count
    // back before the next step and fails loudly) -- never written out of bounds
    __shared__ int s_end[256];        // inclusive scan of the block's tile counts, relative to the block's base
    __shared__ uint32_t s_geo[256];   // x0 | y0 << 16
    __shared__ uint32_t s_w[256];     // rectangle width
    __shared__ uint32_t s_key0[256];  // camera * tiles
    __shared__ int32_t s_pid[256];
    const int t = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * 256;
    const int np = (int)min((int64_t)256, n_pairs - first);
    const int base = first == 0 ? 0 : cum_sorted[first - 1];
    {
        const int64_t sidx = first + min(t, np - 1);
        const uint64_t r = rects_sorted[sidx];
        const int32_t pid = perm[sidx];
        s_end[t] = cum_sorted[sidx] - base;
        s_geo[t] = (uint32_t)(r & 0xFFFFFFFFull);
        s_w[t] = (uint32_t)((r >> 32) & 0xFFFF);
        s_key0[t] = (uint32_t)(pid / N) * (uint32_t)(tile_w * tile_h);
        s_pid[t] = pid;
    }
    __syncthreads();
    const int total = s_end[np - 1];
    for (int o = t; o < total; o += 256) {
        // first pair whose inclusive end exceeds o (pairs without tiles repeat their predecessor's end: skipped)
        int lo = 0, hi = np - 1;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mid = (lo + hi) >> 1;
            const bool right = s_end[mid] <= o;
            lo = right ? mid + 1 : lo;
            hi = right ? hi : mid;
        }
        const int p = lo;
        const int k = o - (p == 0 ? 0 : s_end[p - 1]);   // index inside the pair's rectangle, row major
        const uint32_t w = s_w[p], geo = s_geo[p];
        // k / w with k < w * h <= 2^20 and w < 2^10: float estimate, corrected by one either way
        int q = (int)((float)k * __builtin_amdgcn_rcpf((float)w));
        int rem = k - q * (int)w;
        if (rem < 0) { --q; rem += (int)w; }
        if (rem >= (int)w) { ++q; rem -= (int)w; }
        const uint32_t tx = (geo & 0xFFFF) + (uint32_t)rem, ty = (geo >> 16) + (uint32_t)q;
        // (a true total above 2^31 wraps the int32 scan: a negative position must not pass the capacity test)
        if ((int64_t)base + o >= 0 && (int64_t)base + o < cap) {
            tile_keys[base + o] = s_key0[p] + ty * (uint32_t)tile_w + tx;
            vals[base + o] = s_pid[p];
        }
    }
}

int st3r_isect_emit_rects_impl(hipStream_t s, int N, int C, const int32_t* perm, const int32_t* cum_sorted,
                               const uint64_t* rects_sorted, int tile_w, int tile_h, uint32_t* tile_keys,
                               int32_t* vals, int64_t cap) {
    const int64_t n_pairs = (int64_t)N * C;
    if (n_pairs == 0) return ST3R_OK;
    hipLaunchKernelGGL(k_isect_emit_rects, dim3(ceil_div(n_pairs, 256)), dim3(256), 0, s, N, n_pairs, perm, cum_sorted,
                       rects_sorted, tile_w, tile_h, tile_keys, vals, cap);
    LAUNCH_CHECK();
    return ST3R_OK;
}
