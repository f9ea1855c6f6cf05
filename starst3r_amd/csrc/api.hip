// C-ABI glue: context, scratch arena, error reporting and the fused train / render steps
// that chain the stage kernels on one stream using only ctx scratch.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void st3r_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ST3R_EXPORT int st3r_version(void) { return ST3R_VERSION; }
ST3R_EXPORT const char* st3r_last_error(void) { return g_err; }

ST3R_EXPORT int st3r_ctx_create(int device, st3r_ctx** out) {
    ARG_CHECK(out);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    ARG_CHECK(device >= 0 && device < ndev);
    HIP_TRY(hipSetDevice(device));
    st3r_ctx* c = new (std::nothrow) st3r_ctx();
    if (!c) { st3r_set_error("out of host memory"); return ST3R_ERR_NOMEM; }
    memset(c, 0, sizeof(*c));
    c->device = device;
    hipError_t e = hipHostMalloc((void**)&c->pinned, 64 * sizeof(int64_t), hipHostMallocDefault);
    if (e != hipSuccess) { delete c; st3r_set_error("hipHostMalloc: %s", hipGetErrorString(e)); return ST3R_ERR_HIP; }
    *out = c;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_ctx_destroy(st3r_ctx* ctx) {
    if (!ctx) return ST3R_OK;
    (void)hipSetDevice(ctx->device);
    (void)st3r_comm_destroy(ctx);
    for (int i = 0; i < SLOT_COUNT; ++i)
        if (ctx->slot_ptr[i]) (void)hipFree(ctx->slot_ptr[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->count_event) (void)hipEventDestroy(ctx->count_event);
    if (ctx->peer_event) (void)hipEventDestroy(ctx->peer_event);
    if (ctx->comm_stream) {
        for (int j = 0; j < ST3R_MAX_RANGES; ++j) {
            (void)hipEventDestroy(ctx->ev_range_bwd[j]); (void)hipEventDestroy(ctx->ev_range_red[j]);
        }
        (void)hipStreamDestroy(ctx->comm_stream);
    }
    if (ctx->prof_ev[0][0][0])
        for (int r = 0; r < PROF_RING; ++r)
            for (int st = 0; st < STG_COUNT; ++st)
                for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ctx->prof_ev[r][st][k]);
    delete ctx;
    return ST3R_OK;
}

ST3R_EXPORT int64_t st3r_ctx_arena_bytes(st3r_ctx* ctx) {
    if (!ctx) return 0;
    int64_t t = 0;
    for (int i = 0; i < SLOT_COUNT; ++i) t += (int64_t)ctx->slot_bytes[i];
    return t;
}

int st3r_arena_get(st3r_ctx* ctx, int slot, size_t bytes, void** out) {
    int grown;
    return st3r_arena_get2(ctx, slot, bytes, out, &grown);
}

int st3r_arena_get2(st3r_ctx* ctx, int slot, size_t bytes, void** out, int* grown) {
    *grown = 0;
    if (bytes == 0) bytes = 16;
    if (ctx->slot_bytes[slot] < bytes) {
        // grow with 25% headroom so slowly growing intersection counts do not reallocate every step
        size_t want = bytes + bytes / 4;
        want = (want + 255) & ~(size_t)255;
        if (ctx->slot_ptr[slot]) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(ctx->slot_ptr[slot]));
            ctx->slot_ptr[slot] = nullptr; ctx->slot_bytes[slot] = 0;
        }
        hipError_t e = hipMalloc(&ctx->slot_ptr[slot], want);
        if (e != hipSuccess) {
            st3r_set_error("arena slot %d: hipMalloc(%zu) failed: %s", slot, want, hipGetErrorString(e));
            return ST3R_ERR_NOMEM;
        }
        ctx->slot_bytes[slot] = want;
        *grown = 1;
    }
    *out = ctx->slot_ptr[slot];
    return ST3R_OK;
}

// Debug/test hook: device pointer and capacity of a scratch buffer of the last fused step.
// which: 0 = sorted pair ids ("flatten ids", int32 [n_isects]), 1 = tile offsets (int32 [C*tiles]),
//        2 = splat records (float [C*N*12]), 3 = inclusive tile scan in pair-id order (int32 [C*N])
ST3R_EXPORT int st3r_ctx_peek(st3r_ctx* ctx, void* stream, int which, void* dst, int64_t bytes) {
    ARG_CHECK(ctx && dst && bytes >= 0 && which >= 0 && which <= 10);
    static const int slots[11] = {SLOT_VALS_B, SLOT_OFFSETS, SLOT_SPLATS, SLOT_CUM,
                                  SLOT_MCMC_CUM, SLOT_MCMC_DEAD, SLOT_MCMC_SAMPLED, SLOT_MCMC_COUNT,
                                  SLOT_RGB, SLOT_ALPHA, SLOT_COUNTS};
    ARG_CHECK((size_t)bytes <= ctx->slot_bytes[slots[which]]);
    HIP_TRY(hipMemcpyAsync(dst, ctx->slot_ptr[slots[which]], (size_t)bytes, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return ST3R_OK;
}

// ---- per-stage event timing ----
static const char* k_stage_names[STG_COUNT] = {"project", "scan", "emit", "sort", "offsets", "blend_fwd", "loss",
                                               "blend_bwd", "project_bwd", "adam", "sort_depth"};

ST3R_EXPORT const char* st3r_stage_name(int stage) {
    return (stage >= 0 && stage < STG_COUNT) ? k_stage_names[stage] : "";
}

static void prof_harvest_slot(st3r_ctx* ctx, int slot) {
    for (int st = 0; st < STG_COUNT; ++st) {
        if (!ctx->prof_used[slot][st]) continue;
        float ms = 0.f;
        if (hipEventSynchronize(ctx->prof_ev[slot][st][1]) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->prof_ev[slot][st][0], ctx->prof_ev[slot][st][1]) == hipSuccess) {
            ctx->prof_ms[st] += ms; ctx->prof_n[st] += 1;
        }
        ctx->prof_used[slot][st] = 0;
    }
}

void st3r_prof_begin(st3r_ctx* ctx, hipStream_t s, int stage) {
    if (!((ctx->prof_enabled >> stage) & 1)) return;
    (void)hipEventRecord(ctx->prof_ev[ctx->prof_slot][stage][0], s);
}

void st3r_prof_end(st3r_ctx* ctx, hipStream_t s, int stage) {
    if (!((ctx->prof_enabled >> stage) & 1)) return;
    (void)hipEventRecord(ctx->prof_ev[ctx->prof_slot][stage][1], s);
    ctx->prof_used[ctx->prof_slot][stage] = 1;
}

void st3r_prof_next_step(st3r_ctx* ctx) {
    if (!ctx->prof_enabled) return;
    ctx->prof_slot = (ctx->prof_slot + 1) % PROF_RING;
    prof_harvest_slot(ctx, ctx->prof_slot);  // events from PROF_RING steps ago: long finished
}

ST3R_EXPORT int st3r_ctx_set_debug(st3r_ctx* ctx, int flags) {
    ARG_CHECK(ctx);
    ctx->debug_flags = flags;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_ctx_set_profiling(st3r_ctx* ctx, int enable) {
    ARG_CHECK(ctx && enable >= 0 && enable < 2 + STG_COUNT);
    if (enable && !ctx->prof_ev[0][0][0]) {
        for (int r = 0; r < PROF_RING; ++r)
            for (int st = 0; st < STG_COUNT; ++st)
                for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreate(&ctx->prof_ev[r][st][k]));
    }
    // prof_enabled is the mask of timed stages: enable = 1 -> all of them, enable = 2 + stage -> that stage only
    ctx->prof_enabled = enable == 0 ? 0 : (enable == 1 ? (1 << STG_COUNT) - 1 : (1 << (enable - 2)));
    return ST3R_OK;
}

ST3R_EXPORT int st3r_ctx_get_stage_ms(st3r_ctx* ctx, double* ms_out, int64_t* counts_out) {
    ARG_CHECK(ctx && ms_out && counts_out);
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->prof_ev[0][0][0])
        for (int r = 0; r < PROF_RING; ++r) prof_harvest_slot(ctx, r);
    for (int st = 0; st < STG_COUNT; ++st) {
        ms_out[st] = ctx->prof_ms[st]; counts_out[st] = ctx->prof_n[st];
        ctx->prof_ms[st] = 0; ctx->prof_n[st] = 0;
    }
    return ST3R_OK;
}

// ---- internal stage launchers (other translation units) ----
int st3r_isect_scan_impl(st3r_ctx* ctx, hipStream_t s, int64_t n_pairs, const int32_t* tiles, int32_t* cum,
                         int64_t* n_isects_host, const void* pack_rects, int rect32, uint64_t* pack_out,
                         int32_t** total_dev_out, int32_t* total_copy, int32_t* total_host);
int st3r_isect_emit_impl(hipStream_t s, int N, int C, const float* splats, const int32_t* cum, int tile_size,
                         int tile_w, int tile_h, int64_t* isect_ids, int32_t* flatten_ids);
int st3r_sort_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, int64_t* keys_in, int32_t* vals_in,
                   int64_t* keys_out, int32_t* vals_out);
int st3r_isect_offsets_impl(hipStream_t s, int64_t n_isects, const int64_t* ids, int C, int tile_w, int tile_h,
                            int32_t* offsets);
int st3r_project_impl(st3r_ctx* ctx, hipStream_t s, int N, int C, const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* sh, int sh_stride, const float* viewmats, const float* Ks,
                      const float* campos, int width, int height, int tile_size, float eps2d, float near_plane,
                      float far_plane, float radius_clip, float* splats, int32_t* tiles_per_gauss, double* reg_sums,
                      uint64_t* depth_keys, int32_t* depth_vals, int tight, uint32_t key_base, void* rects, int rect32,
                      int reg_overwrite, double* zero_ptr, int zero_n, uint32_t* krange);
int st3r_sort_depth_seg_impl(st3r_ctx* ctx, hipStream_t s, int64_t N, int C, uint32_t* keys_in, int32_t* vals_in,
                             uint32_t* keys_out, int32_t* vals_out, const uint32_t* krange);
int st3r_isect_emit_chain_impl(st3r_ctx* ctx, hipStream_t s, int N, int C, const int32_t* perm, const void* rects,
                               int rect32, int tile_w, int tile_h, uint32_t* tile_keys, int32_t* vals, int64_t cap);
int st3r_records_prepare_impl(hipStream_t s, int N, int C, const float* splats, int tile_size, int tile_w, int tile_h,
                              int tight, int32_t* tiles, uint64_t* depth_keys, int32_t* depth_vals, uint32_t key_base,
                              void* rects, int rect32);
int st3r_isect_offsets32_impl(hipStream_t s, int64_t n_isects, const uint32_t* keys, int C, int tile_w, int tile_h,
                              int32_t* offsets, const int32_t* n_dev);
int st3r_sort_depth_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint64_t* keys_in, int32_t* vals_in,
                         uint64_t* keys_out, int32_t* vals_out);
int st3r_sort_depth32_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                           uint32_t* keys_out, int32_t* vals_out);
int st3r_sort_tile_impl(st3r_ctx* ctx, hipStream_t s, int64_t n, int end_bit, uint32_t* keys_in, int32_t* vals_in,
                        uint32_t* keys_out, int32_t* vals_out, const int32_t* n_dev);
int st3r_blend_fwd_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                        const float* splats, const int32_t* offsets, const int32_t* flat, int64_t n_isects,
                        float* rgb, float* alpha, int32_t* last_ids, bool for_backward, bool end_in_offsets);
int st3r_blend_bwd_impl(st3r_ctx* ctx, hipStream_t s, int C, int W, int H, int tile_w, int tile_h,
                        const float* splats, const int32_t* offsets, const int32_t* flat, int64_t n_isects,
                        const float* alpha, const int32_t* last_ids, const float* v_rgb, const float* v_alpha,
                        const int32_t* cum, const uint64_t* rects, const uint64_t* rectbase, int tight, int64_t n_pairs,
                        float* v_splats,
                        bool end_in_offsets, st3r_vtile_ref* defer);
int st3r_project_sh_bwd_impl(hipStream_t s, int N, int C, const float* means, const float* quats, const float* scales,
                             const float* opacities, const float* sh, int sh_stride, const float* viewmats,
                             const float* Ks, const float* campos, int width, int height, float eps2d,
                             const float* splats, const float* v_splats, float reg_views, float opac_fac,
                             float scale_fac, float* grads, bool accumulate, int g_begin, int g_end, bool range_major,
                             const st3r_vtile_ref* slots);
int st3r_loss_impl(st3r_ctx* ctx, hipStream_t s, int C, int H, int W, const float* render, const float* gt,
                   float w_l1, float w_ssim, double* sums, float* v_render, bool sums_cleared);

static int bit_length_u32(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

// The previous asynchronous step left its record count in pinned memory behind an event: pick it up (it completed long
// ago), remember it as the sizing hint, and fail loudly if that step ran out of capacity (its records past the
// capacity were dropped, so its gradients were incomplete).
static int settle_pending_count(st3r_ctx* ctx) {
    if (!ctx->count_pending) return ST3R_OK;
    HIP_TRY(hipEventSynchronize(ctx->count_event));
    ctx->count_pending = 0;
    const int64_t n = (int64_t)((int32_t*)(ctx->pinned + 8))[0];
    if (n < 0) {
        ctx->isect_hint = 0;
        ctx->view_chunks = (ctx->view_chunks > 0 ? ctx->view_chunks : 1) * 2;
        st3r_set_error("the previous step produced more than 2^31 tile intersections: its gradients were incomplete -- "
                       "repeat it (st3r_gs_train_fwd_bwd / st3r_gs_train_step now walk the views in %d chunks)",
                       ctx->view_chunks);
        return ST3R_ERR_CAPACITY;
    }
    if (n > ctx->count_cap) {
        ctx->isect_hint = 0;   // the next call takes the synchronous path and sizes its buffers exactly
        st3r_set_error("the previous step produced %lld tile intersections, more than the %lld its buffers were sized "
                       "for from the step before (+25 %%): its gradients were incomplete and st3r_adam_step / "
                       "st3r_gs_train_step did NOT apply them (the update is guarded on the device) -- repeat that step",
                       (long long)n, (long long)ctx->count_cap);
        return ST3R_ERR_CAPACITY;
    }
    ctx->isect_hint = n;
    return ST3R_OK;
}

int st3r_peer_status_settle(st3r_ctx* ctx);   // comm.hip

ST3R_EXPORT int st3r_ctx_settle(st3r_ctx* ctx) {
    ARG_CHECK(ctx);
    int rc = st3r_peer_status_settle(ctx);
    if (rc) return rc;
    return settle_pending_count(ctx);
}

ST3R_EXPORT int st3r_ctx_release_scratch(st3r_ctx* ctx) {
    ARG_CHECK(ctx);
    HIP_TRY(hipDeviceSynchronize());
    const int rc = st3r_ctx_settle(ctx);   // an overflow / a peer's failure of the last asynchronous step is still reported
    for (int i = 0; i < SLOT_COUNT; ++i) {
        if (ctx->slot_ptr[i]) (void)hipFree(ctx->slot_ptr[i]);
        ctx->slot_ptr[i] = nullptr; ctx->slot_bytes[i] = 0;   // (every user of a slot's CONTENTS re-initialises on growth)
    }
    return rc;
}

// The 16 device words next to the fused steps: [0] record count of an asynchronous step (k_adam compares it with the
// step's capacity), [4] status word of an exchanged step (comm.hip).  Zeroed when allocated.
int st3r_counts_buffer(st3r_ctx* ctx, hipStream_t s, int32_t** out) {
    void* p; int grown = 0;
    int rc = st3r_arena_get2(ctx, SLOT_COUNTS, sizeof(int32_t) * 16, &p, &grown);
    if (rc) return rc;
    if (grown) HIP_TRY(hipMemsetAsync(p, 0, ctx->slot_bytes[SLOT_COUNTS], s));
    *out = (int32_t*)p;
    return ST3R_OK;
}

#define GET(slot, type, count, var)                                                          \
    type* var;                                                                               \
    {                                                                                        \
        void* _p;                                                                            \
        int _rc = st3r_arena_get(ctx, slot, sizeof(type) * (size_t)(count), &_p);            \
        if (_rc) return _rc;                                                                 \
        var = (type*)_p;                                                                     \
    }

struct RasterOut {
    float* splats; int32_t* offsets; int32_t* flat; int32_t* cum; const uint64_t* rects; uint64_t* rectbase;
    // n_isects: the slot count (sum of the rectangle areas) = capacity of everything indexed by record or slot;
    // n_records: the records emitted (= n_isects on the synchronous path), -1 while the count stays on the device
    int64_t n_isects, n_records, n_isects_ref, n_visible; int tile_w, tile_h;
};

// project -> scan -> emit -> sort -> offsets, all in ctx scratch
static int rasterize_front(st3r_ctx* ctx, hipStream_t s, int N, int C, const float* means, const float* quats,
                           const float* scales, const float* opacities, const float* sh, int sh_stride,
                           const float* viewmats, const float* Ks, const float* campos, int W, int H,
                           double* reg_sums, int tight, const float* records_in, bool allow_async, RasterOut* o,
                           double* loss_sums = nullptr) {
    // reg_sums is OVERWRITTEN with the projection's sums, and loss_sums[0 .. 2C) (the loss kernel's accumulators, when
    // given) is cleared along the way -- by the projection's reduction launch, not by memsets of their own
    // records_in != NULL: the splat records were projected elsewhere (Gaussian-sharded mode); the projection is
    // replaced by k_records_prepare and the records are used in place
    const int tile = 16;
    const int tile_w = (W + tile - 1) / tile, tile_h = (H + tile - 1) / tile;
    const int64_t n_pairs = (int64_t)N * C;
    float* splats = const_cast<float*>(records_in);
    if (!records_in) {
        GET(SLOT_SPLATS, float, n_pairs * ST3R_SPLAT_STRIDE, own);
        splats = own;
    }
    GET(SLOT_CUM, int32_t, n_pairs, cum);
    GET(SLOT_OFFSETS, int32_t, (int64_t)C * tile_w * tile_h + 1, offsets);   // + the total (closes the last tile)
    // Two-level sort (see gs_isect.hip): pairs by (camera | depth) first, then the emitted records by
    // their 32-bit (camera, tile) key with a stable sort -- the same final order as gsplat's single
    // 64-bit (camera | tile | depth) sort at roughly a quarter of the sort traffic.
    // Level-1 keys.  Up to 8 local views: one 32-bit word, camera (3 bits) | depth bits minus those of the near plane
    // (the reference's near = 0.01 and far = 1e10 span < 2^29 float codes) -- the same order as (camera | depth) at
    // 8 instead of 12 bytes per pair and one radix pass less.  More views: 64-bit (camera << 32 | depth bits).
    const float near_plane = 0.01f, far_plane = 1e10f;
    uint32_t near_bits, far_bits;
    memcpy(&near_bits, &near_plane, 4); memcpy(&far_bits, &far_plane, 4);
    const bool key32 = (C <= 8) && (far_bits - near_bits < 0x1FFFFFFFu);
    const int64_t n_sort = n_pairs;
    GET(SLOT_DKEYS_A, uint64_t, n_sort, dkeys_a);
    GET(SLOT_DKEYS_B, uint64_t, n_sort, dkeys_b);
    GET(SLOT_DVALS_A, int32_t, n_sort, dvals_a);
    GET(SLOT_DVALS_B, int32_t, n_sort, perm);
    // packed tile rectangle of every pair (pair-id order; the tile count of a pair is the area of its rectangle, no
    // array of its own): 32-bit entries for tile grids up to 255 x 255 (tile_rect.h), 64-bit beyond -- and under debug
    // flag 64, whose backward reads the 64-bit form
    const int rect32 = (tile_w <= 255 && tile_h <= 255 && !(ctx->debug_flags & 64)) ? 1 : 0;
    GET(SLOT_RECTS, uint64_t, rect32 ? (n_pairs + 1) / 2 : n_pairs, rects);
    int32_t* counts = nullptr;
    { int rc_ = st3r_counts_buffer(ctx, s, &counts); if (rc_) return rc_; }
    // Round 6: with the projection's own reduction at hand (training calls) the level-1 sort runs per camera SEGMENT on keys
    // biased by the smallest depth code of the call -- three 8-bit passes instead of four whenever the scene's depth codes
    // span less than 2^24 (decided on the device: counts[8..10] = bias, sentinel, passes); debug flag 4 keeps the
    // (camera | depth) keys and their four passes
    uint32_t* const krange = (key32 && reg_sums && !records_in && !(ctx->debug_flags & 4)) ? (uint32_t*)(counts + 8) : nullptr;
    st3r_prof_begin(ctx, s, STG_PROJECT);
    const uint32_t key_base = key32 ? near_bits : 0u;
    int rc = records_in
                 ? st3r_records_prepare_impl(s, N, C, splats, tile, tile_w, tile_h, tight, nullptr, dkeys_a, dvals_a,
                                             key_base, rects, rect32)
                 : st3r_project_impl(ctx, s, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos,
                                     W, H, tile, 0.3f, near_plane, far_plane, 0.0f, splats, nullptr, reg_sums, dkeys_a,
                                     dvals_a, tight, key_base, rects, rect32, 1, loss_sums, 2 * C, krange);
    if (!rc && records_in && loss_sums) HIP_TRY(hipMemsetAsync(loss_sums, 0, sizeof(double) * 2 * (size_t)C, s));
    st3r_prof_end(ctx, s, STG_PROJECT);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_SORT_DEPTH);
    const int cam_bits = bit_length_u32((uint32_t)(C - 1));
    rc = krange ? st3r_sort_depth_seg_impl(ctx, s, N, C, (uint32_t*)dkeys_a, dvals_a, (uint32_t*)dkeys_b, perm, krange)
         : key32 ? st3r_sort_depth32_impl(ctx, s, n_sort, 29 + cam_bits, (uint32_t*)dkeys_a, dvals_a, (uint32_t*)dkeys_b,
                                          perm)
                 : st3r_sort_depth_impl(ctx, s, n_sort, 32 + cam_bits, dkeys_a, dvals_a, dkeys_b, perm);
    st3r_prof_end(ctx, s, STG_SORT_DEPTH);
    if (rc) return rc;
    int64_t n_isects = 0;
    st3r_prof_begin(ctx, s, STG_SCAN);
    // pair-id order scan: slot base of every pair for the backward pass's per-(record, tile) partials
    // (the same launch leaves slot base | rectangle as one word per pair for the backward's staging)
    uint64_t* rectbase = nullptr;
    if (tile_w <= 1023 && tile_h <= 1023 && !(ctx->debug_flags & 64)) {   // 10-bit rectangle fields
        GET(SLOT_RECTBASE, uint64_t, n_pairs, rb);
        rectbase = rb;
    }
    // (its total is the record count; the emit kernel finds the write positions of the depth-ordered records itself)
    // The record count is produced on the device.  Steady state (allow_async and a count from an earlier call): no host
    // round trip -- the buffers are sized from the previous count (+25 %, +1024), every kernel downstream reads the count
    // from device memory (the scan's last workgroup leaves it in the ctx's count word as well), and the count travels to
    // pinned memory behind an event that the NEXT call checks (it also notices, loudly, if this call's count exceeded its
    // capacity).  Otherwise (first call, or the caller wants exact statistics back): copy + synchronise, as in round 1.
    const int32_t* n_dev = nullptr;
    const int64_t sig = ((int64_t)N << 34) ^ ((int64_t)C << 26) ^ ((int64_t)W << 13) ^ (int64_t)H;
    const bool async = allow_async && ctx->isect_hint > 0 && ctx->hint_sig == sig;
    int32_t* total_dev = nullptr;
    rc = st3r_isect_scan_impl(ctx, s, n_pairs, nullptr, cum, nullptr, rects, rect32, rectbase, &total_dev,
                              async ? counts : nullptr, async ? (int32_t*)(ctx->pinned + 8) : nullptr);
    st3r_prof_end(ctx, s, STG_SCAN);
    if (rc) return rc;
    if (async) {
        // (the scan's last workgroup has stored the count into the pinned word itself)
        if (!ctx->count_event) HIP_TRY(hipEventCreateWithFlags(&ctx->count_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->count_event, s));
        n_isects = ctx->isect_hint + ctx->isect_hint / 4 + 1024;   // capacity, not the count
        if (ctx->debug_flags & 8) n_isects = ctx->isect_hint / 2;   // test hook: provoke a capacity overflow
        if (n_isects > 2147483647LL) n_isects = 2147483647LL;
        ctx->count_pending = 1; ctx->count_cap = n_isects;
        o->n_visible = -1; o->n_isects_ref = -1;
    } else {
        HIP_TRY(hipMemcpyAsync(ctx->pinned, total_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (reg_sums) HIP_TRY(hipMemcpyAsync(ctx->pinned + 1, reg_sums + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        n_isects = (int64_t)((int32_t*)ctx->pinned)[0];
        if (n_isects < 0) {   // the tile counts are summed in int32
            st3r_set_error("more than 2^31 tile intersections in one call: split the views over more calls / GPUs");
            return ST3R_SPLIT_VIEWS;   // st3r_gs_train_fwd_bwd retries with the views in chunks; others report invalid
        }
        o->n_visible = reg_sums ? (int64_t)((double*)ctx->pinned)[1] : -1;
        o->n_isects_ref = reg_sums ? (int64_t)((double*)ctx->pinned)[2] : n_isects;
        if (allow_async) { ctx->isect_hint = n_isects; ctx->hint_sig = sig; }
    }
    GET(SLOT_KEYS_A, uint32_t, n_isects, tkeys_a);
    GET(SLOT_KEYS_B, uint32_t, n_isects, tkeys_b);
    GET(SLOT_VALS_A, int32_t, n_isects, vals_a);
    GET(SLOT_VALS_B, int32_t, n_isects, vals_b);
    // the sort and the offsets read the record count from device memory in both paths: the pair-order scan's total
    n_dev = total_dev;
    o->n_records = async ? -1 : n_isects;
    if (n_isects > 0) {
        st3r_prof_begin(ctx, s, STG_EMIT);
        rc = st3r_isect_emit_chain_impl(ctx, s, N, C, perm, rects, rect32, tile_w, tile_h, tkeys_a, vals_a, n_isects);
        st3r_prof_end(ctx, s, STG_EMIT);
        if (rc) return rc;
        const int end_bit = bit_length_u32((uint32_t)((int64_t)C * tile_w * tile_h - 1));
        st3r_prof_begin(ctx, s, STG_SORT);
        rc = st3r_sort_tile_impl(ctx, s, n_isects, end_bit, tkeys_a, vals_a, tkeys_b, vals_b, n_dev);
        st3r_prof_end(ctx, s, STG_SORT);
        if (rc) return rc;
    }
    st3r_prof_begin(ctx, s, STG_OFFSETS);
    rc = st3r_isect_offsets32_impl(s, n_isects, tkeys_b, C, tile_w, tile_h, offsets, n_dev);
    st3r_prof_end(ctx, s, STG_OFFSETS);
    if (rc) return rc;
    o->splats = splats; o->offsets = offsets; o->flat = vals_b; o->cum = cum; o->rects = rect32 ? nullptr : rects;
    o->rectbase = rectbase;
    o->n_isects = n_isects;
    o->tile_w = tile_w; o->tile_h = tile_h;
    return ST3R_OK;
}

__global__ void k_finalize_loss(int C, const double* __restrict__ sums, const double* __restrict__ reg_sums,
                                double inv_px, double inv_cnt, double w_l1, double w_ssim, double reg_views,
                                double opac_k, double scale_k, float* __restrict__ loss_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double loss = 0;
        for (int c = 0; c < C; ++c) loss += w_l1 * sums[2 * c] * inv_px + w_ssim * (1.0 - sums[2 * c + 1] * inv_cnt);
        loss += reg_views * (opac_k * reg_sums[0] + scale_k * reg_sums[1]);
        loss_out[0] = (float)loss;
    }
}

// The views [c0, c0 + C) of one training call: rasterize -> loss -> backward; the parameter gradients are written
// (accumulate = false) or added (later view chunks of the same call).
static int train_views(st3r_ctx* ctx, hipStream_t s, int N, int C, const float* means, const float* quats,
                       const float* scales, const float* opacities, const float* sh, int sh_stride,
                       const float* viewmats, const float* Ks, const float* campos, const float* gt_images, int W, int H,
                       float ssim_fac, float opac_fac, float scale_fac, double* sums, double* reg_sums, bool allow_async,
                       bool accumulate, float* grads, RasterOut* ro_out) {
    const int64_t n_pairs = (int64_t)N * C, n_px = (int64_t)C * H * W;
    RasterOut ro;
    int rc = rasterize_front(ctx, s, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, W, H,
                             reg_sums, 1, nullptr, allow_async, &ro, sums);
    if (rc) return rc;
    const bool eio = true;   // the fused path's offsets table carries the total as its last entry
    GET(SLOT_RGB, float, n_px * 3, rgb);
    GET(SLOT_ALPHA, float, n_px, alpha);
    GET(SLOT_LAST, int32_t, n_px, last);
    GET(SLOT_VRENDER, float, n_px * 3, v_rgb);
    // Round 5: the per-pair sums of the backward's (record, tile) slots are taken inside the projection backward: no
    // 48-byte per-pair gradient records, no k_gather_vtile launch (that kernel serves the stand-alone st3r_gs_blend_bwd and
    // st3r_gs_raster_train)
    float* const v_splats = nullptr;
    st3r_vtile_ref slots_{}; st3r_vtile_ref* const slots = &slots_;
    st3r_prof_begin(ctx, s, STG_BLEND_FWD);
    rc = st3r_blend_fwd_impl(ctx, s, C, W, H, ro.tile_w, ro.tile_h, ro.splats, ro.offsets, ro.flat, ro.n_isects, rgb,
                             alpha, last, true, eio);
    st3r_prof_end(ctx, s, STG_BLEND_FWD);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_LOSS);
    rc = st3r_loss_impl(ctx, s, C, H, W, rgb, gt_images, 1.0f - ssim_fac, ssim_fac, sums, v_rgb, true);
    st3r_prof_end(ctx, s, STG_LOSS);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_BLEND_BWD);
    rc = st3r_blend_bwd_impl(ctx, s, C, W, H, ro.tile_w, ro.tile_h, ro.splats, ro.offsets, ro.flat, ro.n_isects, alpha,
                             last, v_rgb, nullptr, ro.cum, (ctx->debug_flags & 2) ? nullptr : ro.rects,
                             (ctx->debug_flags & 2) ? nullptr : ro.rectbase, 1, n_pairs, v_splats,
                             eio, slots);
    st3r_prof_end(ctx, s, STG_BLEND_BWD);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_PROJECT_BWD);
    // Range-wise exchange (st3r_gs_train_step, comm.hip): one launch per Gaussian range with an event behind each, so that
    // a range's gradients can be reduced while the next range is still being computed.  Only for a whole call in one
    // pass: the later view chunks of a chunked call ADD to every range.
    ctx->ranges_recorded = 0;
    if (ctx->n_ranges > 1 && ctx->comm_stream) {
        // the gradients of a range go to the ctx's staging buffer in range-major order (one contiguous piece per range);
        // Adam reads them from there and leaves them in the caller's buffer in its block layout (comm.hip).  The later
        // view chunks of a chunked call ADD to the staged gradients and record the range events again (the exchange waits
        // for an event's LAST record): whether a rank walks its views in chunks or not, it stages every range and takes
        // part in the same K collectives (round 3 sent the first chunk to the staging buffer and the others to the
        // caller's buffer, and fell back to one all-reduce on that rank only)
        GET(SLOT_GSTAGE, float, (int64_t)23 * N, gstage);
        const int K = ctx->n_ranges;
        for (int j = 0; j < K && !rc; ++j) {
            const int g0 = (int)((int64_t)N * j / K), g1 = (int)((int64_t)N * (j + 1) / K);
            rc = st3r_project_sh_bwd_impl(s, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, W,
                                          H, 0.3f, ro.splats, v_splats, (float)C, opac_fac, scale_fac, gstage, accumulate, g0,
                                          g1, true, slots);
            if (!rc) HIP_TRY(hipEventRecord(ctx->ev_range_bwd[j], s));
        }
        if (!rc) ctx->ranges_recorded = K;
    } else {
        rc = st3r_project_sh_bwd_impl(s, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, W, H,
                                      0.3f, ro.splats, v_splats, (float)C, opac_fac, scale_fac, grads, accumulate, 0, -1,
                                      false, slots);
    }
    st3r_prof_end(ctx, s, STG_PROJECT_BWD);
    *ro_out = ro;
    return rc;
}

ST3R_EXPORT int st3r_gs_train_fwd_bwd(st3r_ctx* ctx, void* stream, int N, int C, const float* means,
                                      const float* quats, const float* scales, const float* opacities,
                                      const float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                      const float* campos, const float* gt_images, int width, int height,
                                      float ssim_fac, float opac_fac, float scale_fac, float* grads,
                                      float* loss_out, int64_t* stats_host) {
    ARG_CHECK(ctx && N > 0 && C > 0 && C <= ST3R_MAX_VIEWS && width > 0 && height > 0 && sh_stride >= 12);
    ARG_CHECK((int64_t)N * C < 2147483647LL);   // pair ids, tile counts and their scans are int32
    ARG_CHECK(means && quats && scales && opacities && sh && viewmats && Ks && campos && gt_images && grads && loss_out);
    hipStream_t s = (hipStream_t)stream;
    const int W = width, H = height;
    GET(SLOT_SMALL, double, 2 * (size_t)C + 12, small);
    double* sums = small;              // [C,2]
    double* reg_sums = small + 2 * C;  // [4]: sum sigmoid(o), sum exp(s), visible pairs, reference intersections
    double* reg_scratch = reg_sums + 4;   // the same of the later view chunks (their regulariser sums are repeats)
    st3r_prof_next_step(ctx);
    int rc = settle_pending_count(ctx);
    if (rc) return rc;
    // More than 2^31 tile intersections in one call (the counts are int32): the views are walked in chunks, each a
    // complete rasterize -> loss -> backward whose parameter gradients add up (the loss is a sum over views,
    // starster/gs.py:149-152).  The chunk count sticks to the context; debug flag 32 starts at two chunks (tests).
    int chunks = ctx->view_chunks > 0 ? ctx->view_chunks : 1;
    if ((ctx->debug_flags & 32) && chunks < 2) chunks = 2;
    if (chunks > C) chunks = C;
    int64_t st_vis = 0, st_is = 0, st_ref = 0;
    for (;;) {
        st_vis = st_is = st_ref = 0;
        bool first = true;
        for (int k = 0; k < chunks && !rc; ++k) {
            const int c0 = (int)((int64_t)k * C / chunks), c1 = (int)((int64_t)(k + 1) * C / chunks);
            if (c1 == c0) continue;
            double* rs = first ? reg_sums : reg_scratch;
            RasterOut ro;
            // exact statistics need the count on the host: a caller that passes stats_host pays the synchronisation;
            // chunked calls size every chunk exactly (the hint of the steady state belongs to one set of views); with a
            // communicator attached every step is sized exactly too -- a capacity overflow would surface on ONE rank only,
            // at its next call, while the other ranks are already waiting in the gradient all-reduce
            rc = train_views(ctx, s, N, c1 - c0, means, quats, scales, opacities, sh, sh_stride, viewmats + 16 * c0,
                             Ks + 9 * c0, campos + 3 * c0, gt_images + (int64_t)c0 * H * W * 3, W, H, ssim_fac, opac_fac,
                             scale_fac, sums + 2 * c0, rs, stats_host == nullptr && chunks == 1 && !ctx->comm, !first, grads,
                             &ro);
            if (!rc) { st_vis += ro.n_visible; st_is += ro.n_records; st_ref += ro.n_isects_ref; }
            first = false;
        }
        if (rc == ST3R_SPLIT_VIEWS && chunks < C) {
            chunks = chunks * 2 < C ? chunks * 2 : C;
            rc = ST3R_OK;
            continue;
        }
        break;
    }
    if (rc == ST3R_SPLIT_VIEWS) rc = ST3R_ERR_INVALID;   // a single view above 2^31 (message set where it was found)
    if (rc) return rc;
    if (chunks > 1) ctx->view_chunks = chunks;
    const int Hi = H - 10, Wi = W - 10;
    const double cnt = (Hi > 0 && Wi > 0) ? (double)Hi * Wi * 3 : 0.0;
    hipLaunchKernelGGL(k_finalize_loss, dim3(1), dim3(64), 0, s, C, sums, reg_sums, 1.0 / ((double)H * W * 3),
                       cnt > 0 ? 1.0 / cnt : 0.0, (double)(1.0f - ssim_fac), (double)ssim_fac, (double)C,
                       (double)opac_fac / N, (double)scale_fac / (3.0 * N), loss_out);
    LAUNCH_CHECK();
    if (stats_host) {
        stats_host[0] = st_vis; stats_host[1] = st_is; stats_host[2] = st3r_ctx_arena_bytes(ctx);
        stats_host[3] = st_ref;   // exact: stats_host selects the synchronous path
    }
    return ST3R_OK;
}

// Gaussian-sharded multi-GPU mode, middle phase: this rank owns C views and received the splat records of ALL
// Gaussians for them (projected by the ranks that own the Gaussians).  Sort, blend, loss, blend backward; the
// per-record gradients go back to the owners, which run the projection backward and Adam on their shard.
ST3R_EXPORT int st3r_gs_raster_train(st3r_ctx* ctx, void* stream, int N, int C, const float* records,
                                     const float* gt_images, int width, int height, float ssim_fac,
                                     float* v_records, float* loss_out, int64_t* stats_host) {
    ARG_CHECK(ctx && N > 0 && C > 0 && width > 0 && height > 0 && records && gt_images && v_records && loss_out);
    ARG_CHECK(C <= ST3R_MAX_VIEWS && (int64_t)N * C < 2147483647LL);
    hipStream_t s = (hipStream_t)stream;
    const int W = width, H = height;
    const int64_t n_pairs = (int64_t)N * C, n_px = (int64_t)C * H * W;
    GET(SLOT_SMALL, double, 2 * (size_t)C + 8, small);
    double* sums = small;
    double* reg_sums = small + 2 * C;
    HIP_TRY(hipMemsetAsync(reg_sums, 0, sizeof(double) * 4, s));  // stays zero: the regularisers belong to the owners
    st3r_prof_next_step(ctx);
    RasterOut ro;
    int rc = settle_pending_count(ctx);
    if (rc) return rc;
    rc = rasterize_front(ctx, s, N, C, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, W, H,
                         nullptr, 1, records, stats_host == nullptr, &ro);
    if (rc == ST3R_SPLIT_VIEWS) rc = ST3R_ERR_INVALID;
    if (rc) return rc;
    GET(SLOT_RGB, float, n_px * 3, rgb);
    GET(SLOT_ALPHA, float, n_px, alpha);
    GET(SLOT_LAST, int32_t, n_px, last);
    GET(SLOT_VRENDER, float, n_px * 3, v_rgb);
    st3r_prof_begin(ctx, s, STG_BLEND_FWD);
    rc = st3r_blend_fwd_impl(ctx, s, C, W, H, ro.tile_w, ro.tile_h, ro.splats, ro.offsets, ro.flat, ro.n_isects, rgb,
                             alpha, last, true, true);
    st3r_prof_end(ctx, s, STG_BLEND_FWD);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_LOSS);
    rc = st3r_loss_impl(ctx, s, C, H, W, rgb, gt_images, 1.0f - ssim_fac, ssim_fac, sums, v_rgb, false);
    st3r_prof_end(ctx, s, STG_LOSS);
    if (rc) return rc;
    st3r_prof_begin(ctx, s, STG_BLEND_BWD);
    rc = st3r_blend_bwd_impl(ctx, s, C, W, H, ro.tile_w, ro.tile_h, ro.splats, ro.offsets, ro.flat, ro.n_isects, alpha,
                             last, v_rgb, nullptr, ro.cum, ro.rects, ro.rectbase, 1, n_pairs, v_records, true, nullptr);
    st3r_prof_end(ctx, s, STG_BLEND_BWD);
    if (rc) return rc;
    const int Hi = H - 10, Wi = W - 10;
    const double cnt = (Hi > 0 && Wi > 0) ? (double)Hi * Wi * 3 : 0.0;
    hipLaunchKernelGGL(k_finalize_loss, dim3(1), dim3(64), 0, s, C, sums, reg_sums, 1.0 / ((double)H * W * 3),
                       cnt > 0 ? 1.0 / cnt : 0.0, (double)(1.0f - ssim_fac), (double)ssim_fac, 0.0, 0.0, 0.0, loss_out);
    LAUNCH_CHECK();
    if (stats_host) {
        stats_host[0] = -1; stats_host[1] = ro.n_records; stats_host[2] = st3r_ctx_arena_bytes(ctx); stats_host[3] = -1;
    }
    return ST3R_OK;
}

ST3R_EXPORT int st3r_gs_render(st3r_ctx* ctx, void* stream, int N, int C, const float* means, const float* quats,
                               const float* scales, const float* opacities, const float* sh, int sh_stride,
                               const float* viewmats, const float* Ks, const float* campos, int width, int height,
                               float* rgb, float* alpha, int64_t* stats_host) {
    ARG_CHECK(ctx && N > 0 && C > 0 && width > 0 && height > 0 && sh_stride >= 12);
    // the projection kernel keeps C * 128 B of camera constants in LDS; pair ids are int32
    ARG_CHECK(C <= ST3R_MAX_VIEWS && (int64_t)N * C < 2147483647LL);
    ARG_CHECK(means && quats && scales && opacities && sh && viewmats && Ks && campos && rgb && alpha);
    hipStream_t s = (hipStream_t)stream;
    RasterOut ro;
    int rc = settle_pending_count(ctx);
    if (rc) return rc;
    rc = rasterize_front(ctx, s, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks, campos, width, height,
                         nullptr, 0, nullptr, false, &ro);
    if (rc == ST3R_SPLIT_VIEWS) rc = ST3R_ERR_INVALID;
    if (rc) return rc;
    GET(SLOT_LAST, int32_t, (int64_t)C * height * width, last);
    rc = st3r_blend_fwd_impl(ctx, s, C, width, height, ro.tile_w, ro.tile_h, ro.splats, ro.offsets, ro.flat,
                             ro.n_isects, rgb, alpha, last, false, true);
    if (rc) return rc;
    if (stats_host) {
        stats_host[0] = -1; stats_host[1] = ro.n_records; stats_host[2] = st3r_ctx_arena_bytes(ctx); stats_host[3] = 0;
    }
    return ST3R_OK;
}
