// Internal helpers shared by the HIP translation units of libst3r_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/st3r.h"

#define ST3R_EXPORT extern "C" __attribute__((visibility("default")))

void st3r_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            st3r_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
            return ST3R_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

#define ARG_CHECK(cond)                                                                    \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            st3r_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond);      \
            return ST3R_ERR_INVALID;                                                       \
        }                                                                                  \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// Grow-only scratch arena.  Slots are named so that steady-state iterations never allocate.
enum ArenaSlot {
    SLOT_SPLATS = 0,
    SLOT_TILES,
    SLOT_CUM,
    SLOT_KEYS_A,
    SLOT_KEYS_B,
    SLOT_VALS_A,
    SLOT_VALS_B,
    SLOT_OFFSETS,
    SLOT_RGB,
    SLOT_ALPHA,
    SLOT_LAST,
    SLOT_VRENDER,
    SLOT_VSPLATS,
    SLOT_SSIM_A,
    SLOT_SORT_TMP,
    SLOT_SCAN_TMP,
    SLOT_SMALL,
    SLOT_COUNT
};

struct st3r_ctx {
    int device;
    void* slot_ptr[SLOT_COUNT];
    size_t slot_bytes[SLOT_COUNT];
    int64_t* pinned;  // small pinned host buffer for read-backs
};

// returns a device pointer with at least `bytes` capacity for `slot` (contents undefined after growth)
int st3r_arena_get(st3r_ctx* ctx, int slot, size_t bytes, void** out);

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

