// K4: stable ascending sort of the (isect key, pair id) stream on bits [0, end_bit).
// Replaces cub::DeviceRadixSort::SortPairs as used by gsplat (starster/gs.py:76).
// The sort itself is the hand-written onesweep radix sort of radix_sort.hip; this file binds it to the four key
// shapes of the pipeline (64-bit gsplat keys of the stage API; level-1 (camera | depth) keys, 32- or 64-bit; level-2
// (camera, tile) keys of the fused two-level path -- see gs_isect.hip).
#include "common.h"
#include "radix_sort.h"

int st3r_sort_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, int64_t* keys_in, int32_t* vals_in,
                   int64_t* keys_out, int32_t* vals_out) {
    if (end_bit > 64) end_bit = 64;
    // keys are non-negative (camera id in the top bits, sign clear): sort them as unsigned
    return st3r_radix_sort_u64(ctx, s, n, 0, end_bit, reinterpret_cast<const uint64_t*>(keys_in), vals_in,
                               reinterpret_cast<uint64_t*>(keys_out), vals_out);
}

// 64-bit (camera | depth) keys of the two-level sort, value = pair id
int st3r_sort_depth_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint64_t* keys_in, int32_t* vals_in,
                         uint64_t* keys_out, int32_t* vals_out) {
    return st3r_radix_sort_u64(ctx, s, n, 0, end_bit, keys_in, vals_in, keys_out, vals_out);
}

// packed 32-bit (camera | depth - near) keys of the level-1 sort (C <= 8)
int st3r_sort_depth32_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                           uint32_t* keys_out, int32_t* vals_out) {
    return st3r_radix_sort_u32(ctx, s, n, 0, end_bit, keys_in, vals_in, keys_out, vals_out);
}

// the same keys WITHOUT the camera bits, one segment of N pairs per camera, biased and sorted in as many 8-bit passes as the
// depth range of the call needs (radix_sort.hip: SEG; krange from the projection's reduction)
int st3r_sort_depth_seg_impl(st3r_ctx* ctx, hipStream_t s, int64_t N, int C, uint32_t* keys_in, int32_t* vals_in,
                             uint32_t* keys_out, int32_t* vals_out, const uint32_t* krange) {
    return st3r_radix_sort_u32_segments(ctx, s, N, C, keys_in, vals_in, keys_out, vals_out, krange);
}

// 32-bit (camera, tile) keys, stable: keeps the depth order established by the first level
int st3r_sort_tile_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                        uint32_t* keys_out, int32_t* vals_out, const int32_t* n_dev) {
    if (n_dev) return st3r_radix_sort_u32_devcount(ctx, s, n, n_dev, 0, end_bit, keys_in, vals_in, keys_out, vals_out);
    return st3r_radix_sort_u32(ctx, s, n, 0, end_bit, keys_in, vals_in, keys_out, vals_out);
}

ST3R_EXPORT int st3r_gs_sort(st3r_ctx* ctx, void* stream, int64_t n_isects, int end_bit, int64_t* isect_ids,
                             int32_t* flatten_ids, int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted) {
    ARG_CHECK(ctx && n_isects >= 0 && end_bit > 0);
    if (n_isects == 0) return ST3R_OK;
    ARG_CHECK(isect_ids && flatten_ids && isect_ids_sorted && flatten_ids_sorted);
    return st3r_sort_impl(ctx, (hipStream_t)stream, n_isects, end_bit, isect_ids, flatten_ids, isect_ids_sorted,
                          flatten_ids_sorted);
}

ST3R_EXPORT int st3r_radix_sort_pairs(st3r_ctx* ctx, void* stream, int key_bytes, int64_t n, int begin_bit, int end_bit,
                                      const void* keys_in, const int32_t* vals_in, void* keys_out, int32_t* vals_out) {
    ARG_CHECK(ctx && n >= 0 && (key_bytes == 4 || key_bytes == 8) && begin_bit >= 0 && end_bit > begin_bit);
    ARG_CHECK(end_bit <= 8 * key_bytes && n < 2147483647LL);
    if (n == 0) return ST3R_OK;
    ARG_CHECK(keys_in && keys_out && ((vals_in == nullptr) == (vals_out == nullptr)));
    if (key_bytes == 4)
        return st3r_radix_sort_u32(ctx, (hipStream_t)stream, n, begin_bit, end_bit, (const uint32_t*)keys_in, vals_in,
                                   (uint32_t*)keys_out, vals_out);
    return st3r_radix_sort_u64(ctx, (hipStream_t)stream, n, begin_bit, end_bit, (const uint64_t*)keys_in, vals_in,
                               (uint64_t*)keys_out, vals_out);
}
