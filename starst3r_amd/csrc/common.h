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
    SLOT_CMASK,
    SLOT_TILE_NB,
    SLOT_VTILE,
    SLOT_DKEYS_A,
    SLOT_DKEYS_B,
    SLOT_DVALS_A,
    SLOT_DVALS_B,
    SLOT_CUM_D,
    SLOT_NN_PART,
    SLOT_MCMC_CUM,
    SLOT_MCMC_DEAD,
    SLOT_MCMC_RANK,
    SLOT_MCMC_COUNT,
    SLOT_MCMC_SAMPLED,
    SLOT_MCMC_BINOMS,
    SLOT_REG_PART,
    SLOT_RECTS,
    SLOT_RECTS_D,
    SLOT_RECTBASE,
    SLOT_COUNTS,
    SLOT_PSTAGE,
    SLOT_GSTAGE,
    SLOT_ALIGN_CTL,   // alignment: the chain cache of k_align_update
    SLOT_PAIR_TOUCH,  // blend backward -> projection backward: per-pair stamp "some slot of this pair was written" (gs_blend.hip)
    SLOT_KRANGE_PART, // projection: per-block smallest / largest level-1 key (gs_project.hip)
    SLOT_SCAN_CHAIN,  // single-pass scans (gs_isect.hip): ticket, totals, one status word per tile
    SLOT_COUNT
};

// Stage ids for the optional per-stage HIP-event timing (st3r_ctx_set_profiling).
enum Stage {
    STG_PROJECT = 0,
    STG_SCAN,
    STG_EMIT,
    STG_SORT,
    STG_OFFSETS,
    STG_BLEND_FWD,
    STG_LOSS,
    STG_BLEND_BWD,
    STG_PROJECT_BWD,
    STG_ADAM,
    STG_SORT_DEPTH,
    STG_COUNT
};
#define PROF_RING 64
#define ST3R_MAX_RANGES 8

#define ST3R_SPLIT_VIEWS 1000   // internal: more than 2^31 tile intersections, the caller may retry with fewer views

// the backward's stamped (record, tile) slots, handed to the kernel that sums them per pair (gs_blend.hip -> gs_project_bwd.hip)
// touch (may be NULL): per pair the stamp of the last backward call in which some (record, tile) of the pair contributed --
// written by k_blend_bwd for scenes of many slots per pair, so that the gather can skip the slots of pairs nobody touched
struct st3r_vtile_ref { const int32_t* cum; const float* vtile; int stamp; unsigned vt_cap; const uint32_t* touch; };

struct st3r_ctx {
    int device;
    void* slot_ptr[SLOT_COUNT];
    size_t slot_bytes[SLOT_COUNT];
    int64_t* pinned;  // small pinned host buffer for read-backs
    // profiling: ring of (start, stop) events per stage; elapsed times are harvested lazily
    int debug_flags;  // st3r_ctx_set_debug: bit 0 = blend forward ignores the per-quadrant relevance test; 1: backward
                      // recomputes the tile rectangles; 2 (4): level-1 sort on (camera | depth) keys in four passes; 3: async capacity halved; 5: training calls start at 2 view chunks;
                      // 6: backward gathers rectangle and slot base separately; 7 (128): training forward on the quadrant
                      // kernel; 9 (512): st3r_gs_render on the cell-list kernel; 11 (2048): under a communicator
                      // st3r_gs_train_step behaves as if this rank's forward / backward had failed (comm.hip)
                      // (round 5's experiment switches -- 4096 masked rectangles, 8192 cell-granular backward, 16384 slot sums
                      // in a kernel of their own -- left the library in round 6: tools/experiments/README.md)
    // ground-truth moments registered by the caller (st3r_ctx_set_gt_moments, loss.hip): caller-owned, valid until cleared
    const float* gtm_gt; const float* gtm_mom; int gtm_c, gtm_h, gtm_w;
    int bwd_stamp;  // generation stamp of the per-(record, tile) partial-gradient slots
    uint32_t scan_gen;   // single-pass scan (gs_isect.hip): generation of its status words
    // record count of the fused steps without a host round trip: sizing hint from the last known count, the read-back
    // still in flight (event), and the capacity the in-flight step was given
    int64_t isect_hint, count_cap;
    int64_t hint_sig;   // (N, C, W, H) the hint belongs to: another workload takes the synchronous path
    int count_pending;
    int view_chunks;    // > 1: the training calls walk their views in this many chunks (2^31 intersections per chunk)
    hipEvent_t count_event;
    void* comm;     // ncclComm_t of the view-sharded job (NULL: single replica)
    int comm_owned, comm_rank, comm_size;
    int exchange;   // ST3R_EXCHANGE_*: the form the gradient exchange of st3r_gs_train_step takes (comm.hip)
    int comm_broken;         // a device-side barrier of the direct form timed out: the ranks lost lockstep (comm.hip)
    int peer_pending;        // the status word of the last exchanged step is still to be looked at (comm.hip)
    hipEvent_t peer_event;
    // range-wise exchange (comm.hip): the projection backward runs once per Gaussian range and leaves an event per
    // range; the ranges' all-reduces run on comm_stream behind those events, Adam per range behind the all-reduces
    hipStream_t comm_stream;
    hipEvent_t ev_range_bwd[ST3R_MAX_RANGES], ev_range_red[ST3R_MAX_RANGES];
    void* xwin;              // direct exchange (comm.hip): this rank's exported buffers and the peers' mapped ones
    int n_ranges;            // > 1: train_views splits the projection backward (set by st3r_gs_train_step for one call)
    int ranges_recorded;     // how many ev_range_bwd the last train_views recorded (0: it ran as one launch)
    int prof_enabled;
    hipEvent_t prof_ev[PROF_RING][STG_COUNT][2];
    unsigned char prof_used[PROF_RING][STG_COUNT];
    int prof_slot;          // ring slot of the step being recorded
    double prof_ms[STG_COUNT];
    int64_t prof_n[STG_COUNT];
};

void st3r_prof_begin(st3r_ctx* ctx, hipStream_t s, int stage);
void st3r_prof_end(st3r_ctx* ctx, hipStream_t s, int stage);
void st3r_prof_next_step(st3r_ctx* ctx);

// returns a device pointer with at least `bytes` capacity for `slot` (contents undefined after growth)
int st3r_arena_get(st3r_ctx* ctx, int slot, size_t bytes, void** out);
// same, *grown = 1 when the slot was (re)allocated by this call
int st3r_arena_get2(st3r_ctx* ctx, int slot, size_t bytes, void** out, int* grown);

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

