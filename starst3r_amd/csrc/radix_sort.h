// Hand-written LSD radix sort of (key, int32 value) pairs for gfx950: one histogram launch for all digits, then one
// "onesweep" launch per 8-bit digit (chained scan with decoupled look-back, Adinets & Merrill 2022, restated here
// for wave64 / 160 KB LDS).  Replaces cub::DeviceRadixSort::SortPairs as reached from gsplat's isect_tiles
// (starster/gs.py:76); stable, ascending, bits [begin_bit, end_bit).
#pragma once
#include "common.h"

// keys_in/vals_in are left untouched; the result lands in keys_out/vals_out.  vals may be NULL (keys only).
int st3r_radix_sort_u32(st3r_ctx* ctx, hipStream_t s, int64_t n, int begin_bit, int end_bit, const uint32_t* keys_in,
                        const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out);
int st3r_radix_sort_u64(st3r_ctx* ctx, hipStream_t s, int64_t n, int begin_bit, int end_bit, const uint64_t* keys_in,
                        const int32_t* vals_in, uint64_t* keys_out, int32_t* vals_out);
// item count in device memory (see radix_sort.hip)
int st3r_radix_sort_u32_devcount(st3r_ctx* ctx, hipStream_t s, int64_t n_cap, const int32_t* n_dev, int begin_bit,
                                 int end_bit, const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out,
                                 int32_t* vals_out);
// n_seg segments of seg_n pairs each (the level-1 sort of the fused training calls: one segment per camera), every segment
// sorted on its own by key - krange[0] (the culled pairs' sentinel 0x1FFFFFFF by krange[1]) in krange[2] 8-bit passes -- all
// three read from device memory (written by the projection's reduction: gs_project.hip)
int st3r_radix_sort_u32_segments(st3r_ctx* ctx, hipStream_t s, int64_t seg_n, int n_seg, const uint32_t* keys_in,
                                 const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out, const uint32_t* krange);
