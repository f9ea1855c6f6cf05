// K4: stable ascending sort of the (isect key, pair id) stream on bits [0, end_bit).
// Replaces cub::DeviceRadixSort::SortPairs as used by gsplat (starster/gs.py:76).
//
// Round-1 implementation: rocPRIM's device radix sort (the ROCm counterpart of the cub call
// the reference reaches); the (tile | depth) structure of the key is exploited by the
// planned two-level sort (DESIGN.md "sort").
#include "common.h"

#include <rocprim/rocprim.hpp>

int st3r_sort_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, int64_t* keys_in, int32_t* vals_in,
                   int64_t* keys_out, int32_t* vals_out) {
    if (n == 0) return ST3R_OK;
    if (end_bit > 64) end_bit = 64;
    size_t tmp_bytes = 0;
    // keys are non-negative (camera id in the top bits, sign clear): sort them as unsigned
    auto* ki = reinterpret_cast<uint64_t*>(keys_in);
    auto* ko = reinterpret_cast<uint64_t*>(keys_out);
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, tmp_bytes, &tmp);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit,
                                      s));
    return ST3R_OK;
}

// 64-bit (camera | depth) keys of the two-level sort, value = pair id
int st3r_sort_depth_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint64_t* keys_in, int32_t* vals_in,
                         uint64_t* keys_out, int32_t* vals_out) {
    if (n == 0) return ST3R_OK;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, tmp_bytes, &tmp);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    return ST3R_OK;
}

// packed 32-bit (camera | depth - near) keys of the level-1 sort (C <= 8)
int st3r_sort_depth32_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                           uint32_t* keys_out, int32_t* vals_out) {
    if (n == 0) return ST3R_OK;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, tmp_bytes, &tmp);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit,
                                      s));
    return ST3R_OK;
}

// 32-bit (camera, tile) keys, stable: keeps the depth order established by the first level
int st3r_sort_tile_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                        uint32_t* keys_out, int32_t* vals_out) {
    if (n == 0) return ST3R_OK;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    void* tmp;
    int rc = st3r_arena_get(ctx, SLOT_SORT_TMP, tmp_bytes, &tmp);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                      (unsigned)end_bit, s));
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_sort(st3r_ctx* ctx, void* stream, int64_t n_isects, int end_bit, int64_t* isect_ids,
                             int32_t* flatten_ids, int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted) {
    ARG_CHECK(ctx && n_isects >= 0 && end_bit > 0);
    if (n_isects == 0) return ST3R_OK;
    ARG_CHECK(isect_ids && flatten_ids && isect_ids_sorted && flatten_ids_sorted);
    return st3r_sort_impl(ctx, (hipStream_t)stream, n_isects, end_bit, isect_ids, flatten_ids, isect_ids_sorted,
                          flatten_ids_sorted);
}
