// the one exchange step of the view-sharded train loop -- a sum all-reduce of the [23N]
// gradient buffer over RCCL (xGMI inside a node) -- and the whole iteration as a single C call.
//
// The reference is single-process (starster/gs.py:143-164 loops over all views on one device); sharding the
// views over one process per GPU is valid because the loss is a plain sum over views (gs.py:149-152).
//
// The exchange itself comes in three forms (ST3R_EXCHANGE = allreduce | ranges | rs_ag; default: allreduce).  All three leave every replica with the same parameters; what differs is what overlaps:
//   allreduce  one ncclAllReduce of the 23N floats on the caller's stream after the whole backward, Adam after it
//              (round 1 / 2).  Nothing overlaps: 92 MB at 1 M Gaussians.
//   ranges     the projection backward, the last kernel of the backward, runs once per Gaussian range (K = 4) with an
//              event behind each launch and writes the range's gradients RANGE-MAJOR into a staging buffer of the ctx
//              (the block layout of the caller's buffer would spread a range over five segments: five small collectives
//              per range -- fewer, larger collectives is the rule on xGMI); a range is then ONE contiguous all-reduce
//              of 23 N / K floats (23 MB at 1 M, K = 4) on a second stream as soon as its event fires, i.e. under the
//              projection backward of the next range; Adam of a range starts when its all-reduce is done, i.e. under
//              the all-reduce of the next, reads the staged gradients and leaves them in the caller's buffer.
//              Exposed communication: the last range plus whatever the 0.2 ms of backward / 0.14 ms of Adam cannot cover.
//   rs_ag      reduce-scatter -> Adam on the rank's 1/w of the buffer (moments m, v are only maintained there: 2 x 92 MB
//              of optimizer state become 2 x 92/w MB of live state) -> all-gather of the updated parameters through a
//              staging buffer in gradient-buffer order -> a copy into the parameter tensors.  The same bytes on the links
//              as an all-reduce (which IS a reduce-scatter + all-gather), but on the point-to-point xGMI mesh the two
//              halves are direct sends -- 7 links x 1/8 of the buffer each -- instead of a ring, and the replicated
//              0.14 ms Adam shrinks to 1/w.  The buffer is cut into w equal contiguous pieces in BUFFER order (Adam is
//              element-wise over the 23N scalars, so a piece need not respect Gaussian boundaries); the < w floats that
//              do not divide are all-reduced and updated by everyone.
// None of this has run on more than one GPU (the builder's and the round-end box have one): tests/test_gpu_multi.py
// spawns one process per visible GPU and checks all three against the single-GPU step as soon as two are visible;
// tests/test_gpu_comm.py runs all three with a one-rank communicator.
//
// RCCL is resolved at run time with dlopen (the copy already mapped into the process, e.g. the one a host
// framework ships, is preferred) so the library keeps loading on hosts without RCCL; only the comm entry
// points then fail, loudly.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {
struct RcclApi {
    void* handle;
    ncclResult_t (*get_unique_id)(ncclUniqueId*);
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*comm_destroy)(ncclComm_t);
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*reduce_scatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*group_start)();
    ncclResult_t (*group_end)();
    const char* (*get_error_string)(ncclResult_t);
};

RcclApi* rccl_api() {
    static RcclApi api = {};
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { st3r_set_error("RCCL not found: %s", dlerror()); return nullptr; }
    api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
    api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
    api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
    api.all_reduce = (decltype(api.all_reduce))dlsym(h, "ncclAllReduce");
    api.reduce_scatter = (decltype(api.reduce_scatter))dlsym(h, "ncclReduceScatter");
    api.all_gather = (decltype(api.all_gather))dlsym(h, "ncclAllGather");
    api.group_start = (decltype(api.group_start))dlsym(h, "ncclGroupStart");
    api.group_end = (decltype(api.group_end))dlsym(h, "ncclGroupEnd");
    api.get_error_string = (decltype(api.get_error_string))dlsym(h, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce || !api.get_error_string ||
        !api.reduce_scatter || !api.all_gather || !api.group_start || !api.group_end) {
        st3r_set_error("RCCL library lacks a required symbol");
        return nullptr;
    }
    api.handle = h;
    return &api;
}
}  // namespace

#define RCCL_TRY(api, expr)                                                                       \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            st3r_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api)->get_error_string(_r)); \
            return ST3R_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)

ST3R_EXPORT int st3r_comm_unique_id(char* id_out) {
    ARG_CHECK(id_out);
    RcclApi* api = rccl_api();
    if (!api) return ST3R_ERR_HIP;
    ncclUniqueId id;
    RCCL_TRY(api, api->get_unique_id(&id));
    memcpy(id_out, id.internal, ST3R_COMM_ID_BYTES);
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_init(st3r_ctx* ctx, int world_size, int rank, const char* id) {
    ARG_CHECK(ctx && id && world_size >= 1 && rank >= 0 && rank < world_size && !ctx->comm);
    RcclApi* api = rccl_api();
    if (!api) return ST3R_ERR_HIP;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, ST3R_COMM_ID_BYTES);
    ncclComm_t comm;
    RCCL_TRY(api, api->comm_init_rank(&comm, world_size, uid, rank));
    ctx->comm = comm; ctx->comm_owned = 1; ctx->comm_rank = rank; ctx->comm_size = world_size;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_attach(st3r_ctx* ctx, void* rccl_comm, int world_size, int rank) {
    ARG_CHECK(ctx && rccl_comm && world_size >= 1 && rank >= 0 && rank < world_size && !ctx->comm);
    if (!rccl_api()) return ST3R_ERR_HIP;
    ctx->comm = rccl_comm; ctx->comm_owned = 0; ctx->comm_rank = rank; ctx->comm_size = world_size;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_destroy(st3r_ctx* ctx) {
    if (!ctx || !ctx->comm) return ST3R_OK;
    RcclApi* api = rccl_api();
    if (api && ctx->comm_owned) (void)api->comm_destroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_owned = 0; ctx->comm_size = 0; ctx->comm_rank = 0;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_world(st3r_ctx* ctx, int* world_size, int* rank) {
    ARG_CHECK(ctx && world_size && rank);
    *world_size = ctx->comm ? ctx->comm_size : 1;
    *rank = ctx->comm ? ctx->comm_rank : 0;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_grad_allreduce(st3r_ctx* ctx, void* stream, float* grads, int64_t count) {
    ARG_CHECK(ctx && count >= 0 && (count == 0 || grads));
    if (!ctx->comm || count == 0) return ST3R_OK;  // no communicator: a single replica owns every view
    RcclApi* api = rccl_api();
    if (!api) return ST3R_ERR_HIP;
    RCCL_TRY(api, api->all_reduce(grads, grads, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm,
                                  (hipStream_t)stream));
    return ST3R_OK;
}

int st3r_adam_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities, float* sh,
                   int sh_stride, const float* grads, float* m, float* v, double lr, double b1, double b2,
                   double eps, int step, const int32_t* count_dev, uint32_t count_cap, int64_t i0, int64_t i1,
                   int64_t g0, int64_t g1, float* pstage, const float* gstage, float* grads_out);
int st3r_params_from_stage_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities,
                                float* sh, int sh_stride, const float* pstage, int64_t i0, int64_t i1, int64_t lim,
                                const int32_t* count_dev, uint32_t count_cap);
void st3r_adam_guard(st3r_ctx* ctx, const int32_t** count_dev, uint32_t* count_cap);

enum { EXCH_ALLREDUCE = 0, EXCH_RANGES = 1, EXCH_RS_AG = 2 };
#define EXCH_RANGES_K 4

static int exchange_mode(const st3r_ctx* ctx) {
    const char* e = getenv("ST3R_EXCHANGE");
    if (e && !strcmp(e, "allreduce")) return EXCH_ALLREDUCE;
    if (e && !strcmp(e, "ranges")) return EXCH_RANGES;
    if (e && !strcmp(e, "rs_ag")) return EXCH_RS_AG;
    // Default: the plain all-reduce.  The exchange follows the LAST kernels of the iteration, so the range-wise form can
    // hide at most the projection backward and Adam (0.05 + 0.14 ms at one view per GPU) behind four collectives'
    // latencies, and rs_ag keeps the moments on the own piece only; neither has been measured on more than one GPU --
    // bench.py --gpus N times all three (per_rank.exchange_forms_ms_per_step).
    return EXCH_ALLREDUCE;
}

static int ensure_comm_stream(st3r_ctx* ctx) {
    if (ctx->comm_stream) return ST3R_OK;
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int j = 0; j < ST3R_MAX_RANGES; ++j) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_range_bwd[j], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_range_red[j], hipEventDisableTiming));
    }
    ctx->comm_stream = st;
    return ST3R_OK;
}

// One whole iteration of starster/gs.py:143-164 for this rank's C views: render -> loss -> backward ->
// (exchange of the gradients when a communicator is attached: see the head of this file) -> Adam.  Asynchronous apart
// from the intersection-count read-back inside the rasterizer.
ST3R_EXPORT int st3r_gs_train_step(st3r_ctx* ctx, void* stream, int N, int C, float* means, float* quats, float* scales,
                                   float* opacities, float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                   const float* campos, const float* gt_images, int width, int height, float ssim_fac,
                                   float opac_fac, float scale_fac, float* grads, float* m, float* v, double lr,
                                   double beta1, double beta2, double eps, int step, float* loss_out,
                                   int64_t* stats_host) {
    ARG_CHECK(ctx && grads && m && v && step >= 1);
    hipStream_t s = (hipStream_t)stream;
    const int mode = ctx->comm ? exchange_mode(ctx) : EXCH_ALLREDUCE;
    RcclApi* api = ctx->comm ? rccl_api() : nullptr;
    if (ctx->comm && !api) return ST3R_ERR_HIP;
    if (mode == EXCH_RANGES) {
        int rc0 = ensure_comm_stream(ctx);
        if (rc0) return rc0;
        ctx->n_ranges = N >= 4096 ? EXCH_RANGES_K : 1;
    }
    int rc = st3r_gs_train_fwd_bwd(ctx, stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks,
                                   campos, gt_images, width, height, ssim_fac, opac_fac, scale_fac, grads, loss_out,
                                   stats_host);
    ctx->n_ranges = 0;
    if (rc) return rc;
    const int64_t total = (int64_t)23 * N;
    const int32_t* count_dev; uint32_t count_cap;
    st3r_adam_guard(ctx, &count_dev, &count_cap);
    if (mode == EXCH_RANGES && ctx->ranges_recorded > 1) {
        // ---- range-wise: all-reduce of range j behind its backward event, Adam of range j behind its all-reduce
        const int K = ctx->ranges_recorded;
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        float* gstage = (float*)ctx->slot_ptr[SLOT_GSTAGE];   // range-major: range j = 23 (g1 - g0) floats at 23 g0
        for (int j = 0; j < K; ++j) {
            const int64_t g0 = (int64_t)N * j / K, g1 = (int64_t)N * (j + 1) / K;
            HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_range_bwd[j], 0));
            RCCL_TRY(api, api->all_reduce(gstage + 23 * g0, gstage + 23 * g0, (size_t)(23 * (g1 - g0)), ncclFloat32, ncclSum,
                                          comm, ctx->comm_stream));
            HIP_TRY(hipEventRecord(ctx->ev_range_red[j], ctx->comm_stream));
        }
        st3r_prof_begin(ctx, s, STG_ADAM);
        for (int j = 0; j < K && !rc; ++j) {
            const int64_t g0 = (int64_t)N * j / K, g1 = (int64_t)N * (j + 1) / K;
            HIP_TRY(hipStreamWaitEvent(s, ctx->ev_range_red[j], 0));
            rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps,
                                step, count_dev, count_cap, -1, -1, g0, g1, nullptr, gstage, grads);
        }
        st3r_prof_end(ctx, s, STG_ADAM);
        return rc;
    }
    if (mode == EXCH_RS_AG) {
        // ---- reduce-scatter -> Adam on the own piece -> all-gather of the parameters
        const int w = ctx->comm_size, r = ctx->comm_rank;
        const int64_t q = total / w, tail0 = q * w;
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        void* ps;
        rc = st3r_arena_get(ctx, SLOT_PSTAGE, sizeof(float) * (size_t)total, &ps);
        if (rc) return rc;
        float* pstage = (float*)ps;
        if (q > 0) RCCL_TRY(api, api->reduce_scatter(grads, grads + r * q, (size_t)q, ncclFloat32, ncclSum, comm, s));
        if (total > tail0) RCCL_TRY(api, api->all_reduce(grads + tail0, grads + tail0, (size_t)(total - tail0), ncclFloat32, ncclSum, comm, s));
        st3r_prof_begin(ctx, s, STG_ADAM);
        rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps, step,
                            count_dev, count_cap, r * q, (r + 1) * q, 0, -1, pstage, nullptr, nullptr);
        if (!rc && total > tail0)   // the remainder: every rank holds its sum and updates it itself
            rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps,
                                step, count_dev, count_cap, tail0, total, 0, -1, nullptr, nullptr, nullptr);
        st3r_prof_end(ctx, s, STG_ADAM);
        if (rc) return rc;
        if (q > 0) {
            RCCL_TRY(api, api->all_gather(pstage + r * q, pstage, (size_t)q, ncclFloat32, comm, s));
            rc = st3r_params_from_stage_impl(s, N, means, quats, scales, opacities, sh, sh_stride, pstage, r * q, (r + 1) * q,
                                             tail0, count_dev, count_cap);
        }
        return rc;
    }
    rc = st3r_grad_allreduce(ctx, stream, grads, total);
    if (rc) return rc;
    return st3r_adam_step(ctx, stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2,
                          eps, step);
}
