// the one exchange step of the view-sharded train loop -- a sum all-reduce of the [23N]
// gradient buffer over RCCL (xGMI inside a node) -- and the whole iteration as a single C call.
//
// The reference is single-process (starster/gs.py:143-164 loops over all views on one device); sharding the
// views over one process per GPU is valid because the loss is a plain sum over views (gs.py:149-152).
//
// The exchange itself comes in three forms, a setting of the ctx (st3r_comm_set_exchange; a communicator starts with the
// form named by the environment variable ST3R_EXCHANGE = allreduce | ranges | rs_ag when it is set -- read ONCE, when the
// communicator is created or attached, never written -- and with the plain all-reduce otherwise).  All three leave every
// replica with the same parameters; what differs is what overlaps:
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
//              as an all-reduce (which IS a reduce-scatter + all-gather); the replicated 0.14 ms Adam shrinks to 1/w.
//              WHICH algorithm RCCL runs for ncclReduceScatter / ncclAllGather on the xGMI mesh is RCCL's choice and
//              unknown here (round 4 claimed "direct sends": unverified -- a one-rank communicator prints no tuning
//              decision, profiles/r5_rccl_one_rank_probe.md; RCCL's default for both is a ring).  The buffer is cut into
//              w equal contiguous pieces in BUFFER order (Adam is element-wise over the 23N scalars, so a piece need
//              not respect Gaussian boundaries); the < w floats that do not divide are all-reduced and updated by everyone.
//   direct     (round 5) the same three steps with NO RCCL call on the data path, i.e. the algorithm is this file's, not
//              a tuner's: every rank exports its gradient buffer, a parameter staging buffer and a row of flags through
//              HIP IPC; per step a rank reads ITS piece of the gradients from all w - 1 peers (w - 1 point-to-point
//              links at once, 1/w of the buffer over each) and adds them in rank order, runs Adam on the piece, and reads
//              the other pieces of the updated parameters from their owners.  One owner per element: replicas are
//              bit-identical by construction.  Two device-side barriers per step (k_xbar: system-scope flags in the
//              peers' exported rows, bounded spin); the step's status word rides on them.  Set-up / tear-down are
//              collective (handles travel through one ncclAllGather).  SURVEY section 5's "direct RS + AG over IPC buffers".
// None of this has run on more than one GPU (the builder's and the round-end box have one): tests/test_gpu_multi.py
// spawns one process per visible GPU and checks all four against the single-GPU step as soon as two are visible; on one
// GPU it runs them with emulated ranks -- `direct` then moves its data through REAL same-device HIP IPC mappings and real
// device-side barriers between processes; tests/test_gpu_comm.py runs all four with a one-rank communicator.
//
// RCCL is resolved at run time with dlopen (the copy already mapped into the process, e.g. the one a host
// framework ships, is preferred) so the library keeps loading on hosts without RCCL; only the comm entry
// points then fail, loudly.
#include <dlfcn.h>

#include <vector>

#include <rccl/rccl.h>

#include <chrono>
#include <thread>

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
    // ST3R_RCCL_LIB: the library to bind instead (a site's own RCCL build; the multi-rank-on-one-GPU shim of the tests)
    void* h = nullptr;
    if (const char* e = getenv("ST3R_RCCL_LIB")) {
        h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        if (!h) { st3r_set_error("ST3R_RCCL_LIB=%s: %s", e, dlerror()); return nullptr; }
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
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

// The form a new communicator starts with.  Default: the plain all-reduce.  The exchange follows the LAST kernels of the
// iteration, so the range-wise form can hide at most the projection backward and Adam (0.05 + 0.14 ms at one view per
// GPU) behind four collectives' latencies, and rs_ag keeps the moments on the own piece only; neither has been measured
// on more than one GPU -- bench.py --gpus N times all three (per_rank.exchange_forms_ms_per_step).
static int exchange_from_env() {
    const char* e = getenv("ST3R_EXCHANGE");
    if (e && !strcmp(e, "ranges")) return ST3R_EXCHANGE_RANGES;
    if (e && !strcmp(e, "rs_ag")) return ST3R_EXCHANGE_RS_AG;
    if (e && !strcmp(e, "direct")) return ST3R_EXCHANGE_DIRECT;
    return ST3R_EXCHANGE_ALLREDUCE;
}

ST3R_EXPORT int st3r_comm_set_exchange(st3r_ctx* ctx, int form) {
    ARG_CHECK(ctx && (form == ST3R_EXCHANGE_ALLREDUCE || form == ST3R_EXCHANGE_RANGES || form == ST3R_EXCHANGE_RS_AG ||
                      form == ST3R_EXCHANGE_DIRECT));
    ctx->exchange = form;
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_get_exchange(st3r_ctx* ctx, int* form) {
    ARG_CHECK(ctx && form);
    *form = ctx->exchange;
    return ST3R_OK;
}

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
    ctx->exchange = exchange_from_env();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_comm_attach(st3r_ctx* ctx, void* rccl_comm, int world_size, int rank) {
    ARG_CHECK(ctx && rccl_comm && world_size >= 1 && rank >= 0 && rank < world_size && !ctx->comm);
    if (!rccl_api()) return ST3R_ERR_HIP;
    ctx->comm = rccl_comm; ctx->comm_owned = 0; ctx->comm_rank = rank; ctx->comm_size = world_size;
    ctx->exchange = exchange_from_env();
    return ST3R_OK;
}

static void xwin_destroy(st3r_ctx* ctx, RcclApi* api);

ST3R_EXPORT int st3r_comm_destroy(st3r_ctx* ctx) {
    if (!ctx || !ctx->comm) return ST3R_OK;
    RcclApi* api = rccl_api();
    if (ctx->xwin) xwin_destroy(ctx, api);   // (collective: see st3r.h)
    if (api && ctx->comm_owned) (void)api->comm_destroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_owned = 0; ctx->comm_size = 0; ctx->comm_rank = 0; ctx->comm_broken = 0;
    ctx->peer_pending = 0;   // nobody is left to repeat a failed step with: a later single-process call must not report it
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
                   double eps, int step, const int32_t* count_dev, uint32_t count_cap, const int32_t* status_dev, int64_t i0,
                   int64_t i1, int64_t g0, int64_t g1, float* pstage, const float* gstage, float* grads_out);
int st3r_params_from_stage_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities,
                                float* sh, int sh_stride, const float* pstage, int64_t i0, int64_t i1, int64_t lim,
                                const int32_t* count_dev, uint32_t count_cap, const int32_t* status_dev);
void st3r_adam_guard(st3r_ctx* ctx, const int32_t** count_dev, uint32_t* count_cap);

#define EXCH_RANGES_K 4

// Under st3r_gs_train_step with ST3R_EXCHANGE_RS_AG a rank maintains the Adam moments of ITS piece of the 23N buffer
// only (w equal contiguous pieces in buffer order; the < w floats that do not divide are maintained by everyone).
// Leaving that form -- e.g. for a refinement run that grows the Gaussian set and so moves the piece boundaries -- needs
// the moments replicated again: in-place all-gather of the ranks' pieces of `buf` (count = 23 N floats).
ST3R_EXPORT int st3r_comm_allgather_pieces(st3r_ctx* ctx, void* stream, float* buf, int64_t count) {
    ARG_CHECK(ctx && count >= 0 && (count == 0 || buf));
    if (!ctx->comm || count == 0) return ST3R_OK;
    RcclApi* api = rccl_api();
    if (!api) return ST3R_ERR_HIP;
    const int64_t q = count / ctx->comm_size;
    if (q > 0)
        RCCL_TRY(api, api->all_gather(buf + (int64_t)ctx->comm_rank * q, buf, (size_t)q, ncclFloat32, (ncclComm_t)ctx->comm,
                                      (hipStream_t)stream));
    return ST3R_OK;
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

// ---- a step that fails on ONE rank must not strand the others inside a collective ----
// Every step under a communicator carries a status word: 1 on a rank whose forward / backward failed (out of memory,
// an invalid argument that only its shard triggers, ...), max-all-reduced together with the gradients.  The failing
// rank still issues every collective of the step -- on whatever its buffers hold -- and then returns its error; on ALL
// ranks the Adam update of the step is skipped on the device (k_adam's guard reads the reduced word), and every other
// rank learns about it at its next training call (or st3r_ctx_settle): ST3R_ERR_PEER, parameters untouched,
// replicas still identical.
int st3r_counts_buffer(st3r_ctx* ctx, hipStream_t s, int32_t** out);   // api.hip: 16 device words, zeroed when allocated
#define PEER_WORD 4           // index of the status word inside the counts buffer
#define PEER_PINNED 24        // its read-back slot in ctx->pinned (int64 units)

int st3r_peer_status_settle(st3r_ctx* ctx) {
    if (!ctx->peer_pending) return ST3R_OK;
    HIP_TRY(hipEventSynchronize(ctx->peer_event));
    ctx->peer_pending = 0;
    const int32_t word = ((volatile int32_t*)(ctx->pinned + PEER_PINNED))[0];
    if (word & 2) {
        // a device-side barrier of the direct form gave up on a peer: the ranks are no longer in lockstep (this rank may
        // have applied a step a peer did not, or hold parameter pieces the peer never delivered) -- nothing further is
        // exchanged over this communicator
        ctx->comm_broken = 1;
        st3r_set_error("direct exchange: a peer did not arrive at a device-side barrier within the time-out "
                       "(ST3R_XBAR_TIMEOUT_MS): the ranks are out of lockstep and the replicas may differ by one step -- "
                       "this communicator is finished (st3r_comm_destroy, restore the parameters, attach a new one)");
        return ST3R_ERR_PEER;
    }
    if (word != 0) {
        st3r_set_error("the previous training step failed on a rank of the communicator: no rank applied its "
                       "update (replicas are unchanged and identical; every rank gets this code and repeats the step) "
                       "-- see the failing rank's own error");
        return ST3R_ERR_PEER;
    }
    return ST3R_OK;
}

// ---- direct exchange: peer windows over HIP IPC ----
#define XW_MAX_RANKS 64
struct XWindow {
    int w, r;
    size_t cap;                                   // floats per data buffer
    float* grad; float* param; unsigned long long* flag;   // this rank's exported buffers (flag: XW_MAX_RANKS words)
    int flag_uncached;
    void* peer[3][XW_MAX_RANKS];                  // [grad | param | flag][rank]: mapped peers, own entries = own buffers
    void** dev_tab;                               // the same three tables on the device: dev_tab + k * XW_MAX_RANKS
    unsigned long long gen;                       // barrier generation (the same on every rank: lockstep)
    long long timeout_ticks;                      // bound of a barrier's spin (100 MHz ticks)
};
struct XHandles { hipIpcMemHandle_t h[3]; };

static void xwin_free_local(XWindow* x) {
    for (int k = 0; k < 3; ++k)
        for (int p = 0; p < x->w; ++p)
            if (p != x->r && x->peer[k][p]) (void)hipIpcCloseMemHandle(x->peer[k][p]);
    if (x->grad) (void)hipFree(x->grad);
    if (x->param) (void)hipFree(x->param);
    if (x->flag) (void)hipFree(x->flag);
    if (x->dev_tab) (void)hipFree(x->dev_tab);
    delete x;
}

// every rank has finished reading its peers (device idle) AND everybody knows it (one tiny all-reduce, synchronised):
// only then may the mappings and the buffers go
static void xwin_destroy(st3r_ctx* ctx, RcclApi* api) {
    XWindow* x = (XWindow*)ctx->xwin;
    ctx->xwin = nullptr;
    (void)hipDeviceSynchronize();
    int32_t* word = nullptr;
    bool peers_done = true;
    if (api && x->w > 1 && hipMalloc((void**)&word, sizeof(int32_t)) == hipSuccess) {
        (void)hipMemset(word, 0, sizeof(int32_t));
        (void)api->all_reduce(word, word, 1, ncclInt32, ncclMax, (ncclComm_t)ctx->comm, nullptr);
        // bounded: a peer that crashed or tears down in another order must not hang this rank's destructor
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, nullptr) == hipSuccess) {
            const char* te = getenv("ST3R_XBAR_TIMEOUT_MS");
            const double limit_s = (te ? atof(te) : 20000.0) * 1e-3;
            const auto t0 = std::chrono::steady_clock::now();
            while (hipEventQuery(ev) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) { peers_done = false; break; }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        } else {
            (void)hipStreamSynchronize(nullptr);
        }
        if (ev && peers_done) (void)hipEventDestroy(ev);
        if (peers_done) (void)hipFree(word);
    }
    // peers that did not answer may still be reading this rank's buffers: the window is then LEAKED rather than unmapped
    // under them (the process is on its way out or the communicator is broken anyway)
    if (peers_done) xwin_free_local(x);
}

// (Re)builds the window for `total` floats per buffer.  COLLECTIVE: all ranks call it in the same step (they hold the
// same Gaussian count, so they decide alike).  The handles travel through one ncclAllGather on device memory.
static int xwin_ensure(st3r_ctx* ctx, RcclApi* api, hipStream_t s, int64_t total) {
    XWindow* x = (XWindow*)ctx->xwin;
    if (x && x->cap >= (size_t)total && x->w == ctx->comm_size) return ST3R_OK;
    if (ctx->comm_size > XW_MAX_RANKS) { st3r_set_error("direct exchange: at most %d ranks", XW_MAX_RANKS); return ST3R_ERR_INVALID; }
    if (x) xwin_destroy(ctx, api);
    HIP_TRY(hipStreamSynchronize(s));
    x = new XWindow();
    memset(x, 0, sizeof(*x));
    x->w = ctx->comm_size; x->r = ctx->comm_rank;
    x->cap = (size_t)total + (size_t)total / 4 + 1024;   // head-room for a growing Gaussian set (the same on every rank)
    const char* te = getenv("ST3R_XBAR_TIMEOUT_MS");
    x->timeout_ticks = (long long)(te ? atof(te) : 20000.0) * 100000LL;   // 100 MHz
#define XW_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            st3r_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            xwin_free_local(x);                                                                   \
            return ST3R_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)
    XW_TRY(hipMalloc((void**)&x->grad, sizeof(float) * x->cap));
    XW_TRY(hipMalloc((void**)&x->param, sizeof(float) * x->cap));
    // The flags are written by peers while this rank's kernel polls them: they MUST live in uncached (fine-grained) memory
    // that the runtime can export.  Ordinary device memory is cached in this device's L2 as non-coherent lines, a
    // system-scope load is not guaranteed to miss there, and a poll could spin on a stale line until the time-out: where
    // such memory cannot be had or exported the form is refused -- the caller stays on an RCCL form.
    x->flag_uncached = 1;
    if (hipExtMallocWithFlags((void**)&x->flag, sizeof(unsigned long long) * XW_MAX_RANKS, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        st3r_set_error("direct exchange: no uncached device memory for the barrier flags (hipExtMallocWithFlags "
                       "hipDeviceMallocUncached failed): use ST3R_EXCHANGE=allreduce | ranges | rs_ag");
        xwin_free_local(x);
        return ST3R_ERR_HIP;
    }
    XW_TRY(hipMemset(x->flag, 0, sizeof(unsigned long long) * XW_MAX_RANKS));
    XHandles mine;
    memset(&mine, 0, sizeof(mine));
    if (x->w > 1) {
        XW_TRY(hipIpcGetMemHandle(&mine.h[0], x->grad));
        XW_TRY(hipIpcGetMemHandle(&mine.h[1], x->param));
        if (hipIpcGetMemHandle(&mine.h[2], x->flag) != hipSuccess) {
            (void)hipGetLastError();
            st3r_set_error("direct exchange: the uncached barrier flags cannot be exported over HIP IPC on this system: "
                           "use ST3R_EXCHANGE=allreduce | ranges | rs_ag");
            xwin_free_local(x);
            return ST3R_ERR_HIP;
        }
    }
    x->peer[0][x->r] = x->grad; x->peer[1][x->r] = x->param; x->peer[2][x->r] = x->flag;
    if (x->w > 1) {
        static_assert(sizeof(XHandles) % 4 == 0, "handles travel as int32 words");
        char* hb = nullptr;
        XW_TRY(hipMalloc((void**)&hb, sizeof(XHandles) * x->w));
#define XW_TRY_HB(expr) do { hipError_t _e2 = (expr); if (_e2 != hipSuccess) { (void)hipFree(hb); XW_TRY(_e2); } } while (0)
        XW_TRY_HB(hipMemcpy(hb + sizeof(XHandles) * x->r, &mine, sizeof(XHandles), hipMemcpyHostToDevice));
        ncclResult_t nr = api->all_gather(hb + sizeof(XHandles) * x->r, hb, sizeof(XHandles) / 4, ncclInt32,
                                          (ncclComm_t)ctx->comm, s);
        if (nr != ncclSuccess) {
            st3r_set_error("direct exchange: ncclAllGather of the IPC handles -> %s", api->get_error_string(nr));
            (void)hipFree(hb); xwin_free_local(x);
            return ST3R_ERR_HIP;
        }
        std::vector<XHandles> all(x->w);
        XW_TRY_HB(hipMemcpyAsync(all.data(), hb, sizeof(XHandles) * x->w, hipMemcpyDeviceToHost, s));
        XW_TRY_HB(hipStreamSynchronize(s));
#undef XW_TRY_HB
        (void)hipFree(hb);
        for (int p = 0; p < x->w; ++p) {
            if (p == x->r) continue;
            for (int k = 0; k < 3; ++k) XW_TRY(hipIpcOpenMemHandle(&x->peer[k][p], all[p].h[k], hipIpcMemLazyEnablePeerAccess));
        }
    }
    XW_TRY(hipMalloc((void**)&x->dev_tab, sizeof(void*) * 3 * XW_MAX_RANKS));
    XW_TRY(hipMemcpy(x->dev_tab, x->peer, sizeof(void*) * 3 * XW_MAX_RANKS, hipMemcpyHostToDevice));
#undef XW_TRY
    ctx->xwin = x;
    return ST3R_OK;
}

// Barrier between the ranks on the device: thread p tells peer p "rank r has arrived at `gen`" (one system-scope store into
// the peer's exported row, the step's failure bit in the low bits) and waits for peer p's word in this rank's own row.
// A peer can be at most ONE barrier ahead (it cannot leave barrier gen + 1 before this rank arrives there), and both
// barriers of a step carry the step's failure bit, so whichever word is read tells the truth about the step.  The wait is
// bounded: a peer that does not arrive within the time-out marks the step failed (2) instead of hanging the device.
// status_out (first barrier of a step only): the OR over the ranks -- the same word the other forms max-all-reduce.
__global__ __launch_bounds__(XW_MAX_RANKS) void k_xbar(unsigned long long* const* __restrict__ flag_tab, int w, int r,
                                                       unsigned long long gen, int my_fail, long long timeout_ticks,
                                                       int32_t* __restrict__ status_out, int accumulate) {
    __shared__ int s_any;
    const int p = threadIdx.x;
    if (p == 0) s_any = my_fail ? 1 : 0;
    __syncthreads();
    if (p < w && p != r) {
        __hip_atomic_store(flag_tab[p] + r, (gen << 2) | (my_fail ? 1ull : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        unsigned long long v;
        for (;;) {
            v = __hip_atomic_load(flag_tab[r] + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((v >> 2) >= gen) break;
            if (wall_clock64() - t0 > timeout_ticks) { v = 2ull; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (v & 3ull) atomicOr(&s_any, (int)(v & 3ull));
    }
    __syncthreads();
    // (second barrier of a step: ORed into the step's word, so that what follows it -- the parameter pieces read from the
    // peers -- and the host's read-back see a time-out of THIS barrier as well)
    if (p == 0 && status_out) *status_out = accumulate ? (*status_out | s_any) : s_any;
    __threadfence_system();
}

// out[i] = sum over the ranks, in rank order, of their exported gradient buffers, for i in [i0, i1) and [t0, t1).
// A peer's buffer is ordinary (coarse-grained) device memory of ANOTHER device, whose lines this device may have cached
// during the previous step: peers are read with system-scope loads (they bypass the non-coherent caches), the own buffer
// with plain ones.  (The barrier kernel's acquire invalidates the L2 of the one XCD it runs on, not the other seven.)
__device__ __forceinline__ float peer_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__global__ __launch_bounds__(256) void k_xreduce(const float* const* __restrict__ grad_tab, int w, int r, int64_t i0,
                                                 int64_t i1, int64_t t0, int64_t t1, float* __restrict__ out) {
    const int64_t n = (i1 - i0) + (t1 - t0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const int64_t i = j < i1 - i0 ? i0 + j : t0 + (j - (i1 - i0));
        float acc = 0.f;
        for (int p = 0; p < w; ++p) {
            const float v = p == r ? grad_tab[p][i] : peer_load(grad_tab[p] + i);
            acc = p == 0 ? v : acc + v;
        }
        out[i] = acc;
    }
}

int st3r_params_from_peers_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities,
                                float* sh, int sh_stride, const float* const* tab, int r, int64_t q, int64_t lim,
                                const int32_t* status_dev);

// One whole iteration of starster/gs.py:143-164 for this rank's C views: render -> loss -> backward ->
// (exchange of the gradients when a communicator is attached: see the head of this file) -> Adam.  Asynchronous apart
// from the intersection-count read-back inside the rasterizer.
ST3R_EXPORT int st3r_gs_train_step(st3r_ctx* ctx, void* stream, int N, int C, float* means, float* quats, float* scales,
                                   float* opacities, float* sh, int sh_stride, const float* viewmats, const float* Ks,
                                   const float* campos, const float* gt_images, int width, int height, float ssim_fac,
                                   float opac_fac, float scale_fac, float* grads, float* m, float* v, double lr,
                                   double beta1, double beta2, double eps, int step, float* loss_out,
                                   int64_t* stats_host) {
    ARG_CHECK(ctx && grads && m && v && step >= 1 && N > 0);
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->comm) {   // a single replica: no exchange
        int rc = st3r_gs_train_fwd_bwd(ctx, stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats, Ks,
                                       campos, gt_images, width, height, ssim_fac, opac_fac, scale_fac, grads, loss_out,
                                       stats_host);
        if (rc) return rc;
        return st3r_adam_step(ctx, stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2,
                              eps, step);
    }
    RcclApi* api = rccl_api();
    if (!api) return ST3R_ERR_HIP;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    int rc = st3r_peer_status_settle(ctx);   // did the previous step fail somewhere else?
    if (rc) return rc;
    if (ctx->comm_broken) {
        st3r_set_error("this communicator lost lockstep in an earlier step (a device-side barrier timed out): no further "
                       "step is exchanged over it");
        return ST3R_ERR_PEER;
    }
    const int mode = ctx->exchange;
    int32_t* counts = nullptr;
    // (failures from here to the collectives below are failures of the machinery the protocol itself needs: they are
    // returned at once -- nothing can be promised to the other ranks without a stream, an event or 64 bytes of memory)
    rc = st3r_counts_buffer(ctx, s, &counts);
    if (rc) return rc;
    if (!ctx->peer_event) HIP_TRY(hipEventCreateWithFlags(&ctx->peer_event, hipEventDisableTiming));
    const int K = (mode == ST3R_EXCHANGE_RANGES && N >= 4096) ? EXCH_RANGES_K : 1;   // (a function of N alone: rank-consistent)
    if (mode == ST3R_EXCHANGE_RANGES) {
        rc = ensure_comm_stream(ctx);
        if (rc) return rc;
        ctx->n_ranges = K;
    }
    // direct form: the window is (re)built first -- collectively, every rank sees the same N -- and the backward leaves
    // this rank's gradients in its EXPORTED buffer (the caller's buffer receives the reduced piece)
    XWindow* xw = nullptr;
    if (mode == ST3R_EXCHANGE_DIRECT) {
        rc = xwin_ensure(ctx, api, s, (int64_t)23 * N);
        if (rc) return rc;
        xw = (XWindow*)ctx->xwin;
    }
    float* const grads_local = xw ? xw->grad : grads;
    int rc_local = (ctx->debug_flags & 2048)   // test hook: this rank "fails" before it has computed anything
                       ? (st3r_set_error("debug flag 2048: simulated failure of this rank's step"), ST3R_ERR_NOMEM)
                       : st3r_gs_train_fwd_bwd(ctx, stream, N, C, means, quats, scales, opacities, sh, sh_stride, viewmats,
                                               Ks, campos, gt_images, width, height, ssim_fac, opac_fac, scale_fac,
                                               grads_local, loss_out, stats_host);
    ctx->n_ranges = 0;
    char first_error[512];
    if (rc_local) snprintf(first_error, sizeof(first_error), "%s", st3r_last_error());
    // From here on nothing returns early: a HIP / RCCL call of the exchange machinery that fails is remembered (the first
    // one) and every remaining collective of the step is still issued -- an open ncclGroupStart is always closed, the
    // other ranks are never left waiting for a launch this rank skipped.  Such a failure comes after the
    // status word has travelled, so the other ranks cannot be told: this rank returns the error, see st3r.h.
    int rc_mach = ST3R_OK;
    char mach_error[512];
#define SOFT_FAIL(what, why)                                                                                    \
    do {                                                                                                        \
        if (!rc_mach) { rc_mach = ST3R_ERR_HIP; snprintf(mach_error, sizeof(mach_error), "%s:%d: %s -> %s", __FILE__, __LINE__, what, why); } \
    } while (0)
#define SOFT_RCCL(expr)                                                       \
    do {                                                                      \
        ncclResult_t _r = (expr);                                             \
        if (_r != ncclSuccess) SOFT_FAIL(#expr, api->get_error_string(_r));   \
    } while (0)
#define SOFT_HIP(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) SOFT_FAIL(#expr, hipGetErrorString(_e));        \
    } while (0)
    // ---- the status word, reduced with the gradients
    const int64_t total = (int64_t)23 * N;
    if (xw) {   // direct form: the first barrier of the step carries the status word
        ++xw->gen;
        hipLaunchKernelGGL(k_xbar, dim3(1), dim3(XW_MAX_RANKS), 0, s, (unsigned long long* const*)(xw->dev_tab + 2 * XW_MAX_RANKS),
                           xw->w, xw->r, xw->gen, rc_local ? 1 : 0, xw->timeout_ticks, counts + PEER_WORD, 0);
        SOFT_HIP(hipGetLastError());
    } else {
        SOFT_HIP(hipMemsetAsync(counts + PEER_WORD, rc_local ? 1 : 0, sizeof(int32_t), s));
    }
    // (plain all-reduce form: the status word and the gradients travel as ONE grouped launch)
    const bool grouped = mode == ST3R_EXCHANGE_ALLREDUCE || (K == 1 && mode == ST3R_EXCHANGE_RANGES);
    if (grouped) SOFT_RCCL(api->group_start());
    if (!xw) SOFT_RCCL(api->all_reduce(counts + PEER_WORD, counts + PEER_WORD, 1, ncclInt32, ncclMax, comm, s));
    if (grouped) {
        SOFT_RCCL(api->all_reduce(grads, grads, (size_t)total, ncclFloat32, ncclSum, comm, s));
        SOFT_RCCL(api->group_end());
    }
    SOFT_HIP(hipMemcpyAsync((int32_t*)(ctx->pinned + PEER_PINNED), counts + PEER_WORD, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SOFT_HIP(hipEventRecord(ctx->peer_event, s));
    ctx->peer_pending = 1;
    const int32_t* guard; uint32_t count_cap;
    st3r_adam_guard(ctx, &guard, &count_cap);
    const int32_t* status = counts + PEER_WORD;   // the reduced status word: read by THIS step's update kernels only
    rc = ST3R_OK;
    if (mode == ST3R_EXCHANGE_RANGES && K > 1) {
        // ---- range-wise: all-reduce of range j behind its backward event, Adam of range j behind its all-reduce.
        // (A rank whose step failed before the staging buffer existed sends its gradient buffer instead -- the sums are
        // discarded anyway --; its range events were never recorded, which hipStreamWaitEvent treats as complete.)
        float* gstage = ctx->ranges_recorded == K ? (float*)ctx->slot_ptr[SLOT_GSTAGE] : nullptr;
        float* sendbuf = gstage ? gstage : grads;
        for (int j = 0; j < K; ++j) {
            const int64_t g0 = (int64_t)N * j / K, g1 = (int64_t)N * (j + 1) / K;
            SOFT_HIP(hipStreamWaitEvent(ctx->comm_stream, gstage ? ctx->ev_range_bwd[j] : ctx->peer_event, 0));
            SOFT_RCCL(api->all_reduce(sendbuf + 23 * g0, sendbuf + 23 * g0, (size_t)(23 * (g1 - g0)), ncclFloat32, ncclSum,
                                          comm, ctx->comm_stream));
            SOFT_HIP(hipEventRecord(ctx->ev_range_red[j], ctx->comm_stream));
        }
        st3r_prof_begin(ctx, s, STG_ADAM);
        for (int j = 0; j < K; ++j) {
            const int64_t g0 = (int64_t)N * j / K, g1 = (int64_t)N * (j + 1) / K;
            SOFT_HIP(hipStreamWaitEvent(s, ctx->ev_range_red[j], 0));
            if (gstage && !rc)
                rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps,
                                    step, guard, count_cap, status, -1, -1, g0, g1, nullptr, gstage, grads);
        }
        st3r_prof_end(ctx, s, STG_ADAM);
    } else if (mode == ST3R_EXCHANGE_RS_AG) {
        // ---- reduce-scatter -> Adam on the own piece -> all-gather of the parameters
        const int w = ctx->comm_size, r = ctx->comm_rank;
        const int64_t q = total / w, tail0 = q * w;
        void* ps = nullptr;
        const int rc_ps = st3r_arena_get(ctx, SLOT_PSTAGE, sizeof(float) * (size_t)total, &ps);
        float* pstage = rc_ps ? nullptr : (float*)ps;
        if (rc_ps && !rc_local) { rc_local = rc_ps; snprintf(first_error, sizeof(first_error), "%s", st3r_last_error()); }
        if (q > 0) SOFT_RCCL(api->reduce_scatter(grads, grads + r * q, (size_t)q, ncclFloat32, ncclSum, comm, s));
        if (total > tail0) SOFT_RCCL(api->all_reduce(grads + tail0, grads + tail0, (size_t)(total - tail0), ncclFloat32, ncclSum, comm, s));
        st3r_prof_begin(ctx, s, STG_ADAM);
        if (pstage) {
            rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps, step,
                                guard, count_cap, status, r * q, (r + 1) * q, 0, -1, pstage, nullptr, nullptr);
            if (!rc && total > tail0)   // the remainder: every rank holds its sum and updates it itself
                rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps,
                                    step, guard, count_cap, status, tail0, total, 0, -1, nullptr, nullptr, nullptr);
        }
        st3r_prof_end(ctx, s, STG_ADAM);
        if (q > 0) {
            // (without a staging buffer -- this rank is failing -- the gradient buffer stands in: the others' pieces are
            // discarded everywhere, the status word says so)
            float* ag = pstage ? pstage : grads;
            SOFT_RCCL(api->all_gather(ag + r * q, ag, (size_t)q, ncclFloat32, comm, s));
            if (pstage && !rc)
                rc = st3r_params_from_stage_impl(s, N, means, quats, scales, opacities, sh, sh_stride, pstage, r * q,
                                                 (r + 1) * q, tail0, guard, count_cap, status);
        }
    } else if (mode == ST3R_EXCHANGE_DIRECT) {
        // ---- own piece summed from the peers' exported buffers -> Adam on it -> barrier -> the other pieces of the
        // parameters from their owners' staging buffers
        const int w = xw->w, r = xw->r;
        const int64_t q = total / w, tail0 = q * w;
        int blocks = ceil_div(q + (total - tail0), 256 * 4);
        if (blocks < 1) blocks = 1;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_xreduce, dim3(blocks), dim3(256), 0, s, (const float* const*)xw->dev_tab, w, r, r * q,
                           (r + 1) * q, tail0, total, grads);
        SOFT_HIP(hipGetLastError());
        st3r_prof_begin(ctx, s, STG_ADAM);
        if (q > 0)
            rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps, step,
                                guard, count_cap, status, r * q, (r + 1) * q, 0, -1, xw->param, nullptr, nullptr);
        if (!rc && total > tail0)   // the remainder: every rank summed it itself and updates it itself
            rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps,
                                step, guard, count_cap, status, tail0, total, 0, -1, nullptr, nullptr, nullptr);
        st3r_prof_end(ctx, s, STG_ADAM);
        ++xw->gen;
        hipLaunchKernelGGL(k_xbar, dim3(1), dim3(XW_MAX_RANKS), 0, s, (unsigned long long* const*)(xw->dev_tab + 2 * XW_MAX_RANKS),
                           w, r, xw->gen, rc_local ? 1 : 0, xw->timeout_ticks, counts + PEER_WORD, 1);
        SOFT_HIP(hipGetLastError());
        // the word once more for the host, now with the second barrier in it (the event's last record is what the next
        // call waits for)
        SOFT_HIP(hipMemcpyAsync((int32_t*)(ctx->pinned + PEER_PINNED), counts + PEER_WORD, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SOFT_HIP(hipEventRecord(ctx->peer_event, s));
        if (!rc)
            rc = st3r_params_from_peers_impl(s, N, means, quats, scales, opacities, sh, sh_stride,
                                             (const float* const*)(xw->dev_tab + XW_MAX_RANKS), r, q, tail0, status);
    } else {   // (the gradients were all-reduced together with the status word above)
        st3r_prof_begin(ctx, s, STG_ADAM);
        rc = st3r_adam_impl(s, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1, beta2, eps, step,
                            guard, count_cap, status, -1, -1, 0, -1, nullptr, nullptr, nullptr);
        st3r_prof_end(ctx, s, STG_ADAM);
    }
    if (rc_local) { st3r_set_error("%s", first_error); return rc_local; }
    if (rc_mach) { st3r_set_error("%s", mach_error); return rc_mach; }
    return rc;
#undef SOFT_HIP
#undef SOFT_RCCL
#undef SOFT_FAIL
}
