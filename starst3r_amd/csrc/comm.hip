// the one exchange step of the view-sharded train loop -- a sum all-reduce of the [23N]
// gradient buffer over RCCL (xGMI inside a node) -- and the whole iteration as a single C call.
//
// The reference is single-process (starster/gs.py:143-164 loops over all views on one device); sharding the
// views over one process per GPU is valid because the loss is a plain sum over views (gs.py:149-152).
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
    api.get_error_string = (decltype(api.get_error_string))dlsym(h, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce || !api.get_error_string) {
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

// One whole iteration of starster/gs.py:143-164 for this rank's C views: render -> loss -> backward ->
// (sum all-reduce of the gradients when a communicator is attached) -> Adam.  Asynchronous apart from the
// intersection-count read-back inside the rasterizer.
ST3R_EXPORT int st3r_gs_train_step(st3r_ctx* ctx, void* stream, int N, int C, float* means, float* quats, float* scales,
                                   float* opacities, float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                   const float* campos, const float* gt_images, int width, int height, float ssim_fac,
                                   float opac_fac, float scale_fac, float* grads, float* m, float* v, double lr,
                                   double beta1, double beta2, double eps, int step, float* loss_out,
                                   int64_t* stats_host) {
    ARG_CHECK(grads && m && v);
    int rc = st3r_gs_train_fwd_bwd(ctx, stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks,
                                   campos, gt_images, width, height, ssim_fac, opac_fac, scale_fac, grads, loss_out,
                                   stats_host);
    if (rc) return rc;
    rc = st3r_grad_allreduce(ctx, stream, grads, (int64_t)23 * N);
    if (rc) return rc;
    return st3r_adam_step(ctx, stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2,
                          eps, step);
}
