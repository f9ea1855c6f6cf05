// TEST INFRASTRUCTURE, not product code: a stand-in for the six RCCL entry points csrc/comm.hip binds, for several
// processes that share ONE GPU (RCCL itself refuses two ranks on one device, and the builder's box has one).  The ranks meet
// in a POSIX shared-memory segment named after the unique id: every collective synchronises the caller's stream, copies the
// send buffer to the host, meets the other ranks at a barrier, combines in RANK ORDER (so every rank gets the same bits)
// and copies the result back.  Slow and synchronous on purpose; what it exercises is the library's control flow with more
// than one rank -- which collectives, on which pieces, in which order -- and the real reduction results.
// Selected with ST3R_RCCL_LIB=<path of this .so> (tests/test_gpu_multi.py).  Types and enum values: rccl/rccl.h.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t; // ncclInt32 = 2, ncclFloat32 = 7
typedef int ncclRedOp_t;    // ncclSum = 0, ncclMax = 2
}

namespace {
constexpr size_t SLOT_BYTES = 512u << 20;  // per rank (sparse: only what a collective touches is ever backed by /dev/shm)
constexpr double BARRIER_TIMEOUT_S = 120.0; // a rank that died must fail the others, not hang them
struct Shared {
    std::atomic<int> arrived, sense, attached;
    int nranks;
    char pad[240];
};
struct Comm {
    Shared* sh; char* slots; int nranks, rank, local_sense; size_t map_bytes; char name[80];
};

double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
bool barrier(Comm* c) {
    c->local_sense ^= 1;
    if (c->sh->arrived.fetch_add(1) == c->nranks - 1) {
        c->sh->arrived.store(0);
        c->sh->sense.store(c->local_sense);
        return true;
    }
    const double t0 = now_s();
    for (unsigned spin = 0; c->sh->sense.load() != c->local_sense; ++spin) {
        sched_yield();
        if ((spin & 4095) == 4095 && now_s() - t0 > BARRIER_TIMEOUT_S) return false;
    }
    return true;
}
size_t elem_size(int dt) { return (dt == 2 || dt == 7 || dt == 3) ? 4 : (dt == 8 ? 8 : 0); }

template <typename T> void combine(T* dst, const T* src, size_t n, int op, bool first) {
    if (first) { memcpy(dst, src, n * sizeof(T)); return; }
    if (op == 0) for (size_t i = 0; i < n; ++i) dst[i] += src[i];
    else if (op == 2) for (size_t i = 0; i < n; ++i) dst[i] = dst[i] > src[i] ? dst[i] : src[i];
}
int reduce_ranks(Comm* c, void* dst, size_t offset_elems, size_t n, int dt, int op) {
    for (int r = 0; r < c->nranks; ++r) {
        const char* src = c->slots + (size_t)r * SLOT_BYTES + offset_elems * 4;
        if (dt == 7) combine((float*)dst, (const float*)src, n, op, r == 0);
        else if (dt == 2) combine((int32_t*)dst, (const int32_t*)src, n, op, r == 0);
        else return 1;
    }
    return 0;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/st3r_fake_rccl_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    c->nranks = nranks; c->rank = rank; c->local_sense = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = sizeof(Shared) + (size_t)nranks * SLOT_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) return 2;   // (fresh segments are zero-filled: counters start at 0)
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->sh = (Shared*)p; c->slots = (char*)p + sizeof(Shared);
    c->sh->attached.fetch_add(1);
    const double t0 = now_s();
    while (c->sh->attached.load() < nranks) {                // collective, like the real call
        sched_yield();
        if (now_s() - t0 > BARRIER_TIMEOUT_S) return 4;
    }
    *out = c;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    (void)barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void*)c->sh, c->map_bytes);
    delete c;
    return 0;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * elem_size(dt);
    if (!elem_size(dt) || bytes > SLOT_BYTES) return 3;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 4;
    std::vector<char> res(bytes);
    if (reduce_ranks(c, res.data(), 0, count, dt, op)) return 3;
    if (!barrier(c)) return 4;
    return hipMemcpy(recv, res.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t dt, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    const size_t bytes = recvcount * c->nranks * elem_size(dt);
    if (elem_size(dt) != 4 || bytes > SLOT_BYTES) return 3;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 4;
    std::vector<char> res(recvcount * 4);
    if (reduce_ranks(c, res.data(), (size_t)c->rank * recvcount, recvcount, dt, op)) return 3;
    if (!barrier(c)) return 4;
    return hipMemcpy(recv, res.data(), recvcount * 4, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    const size_t bytes = sendcount * elem_size(dt);
    if (!elem_size(dt) || bytes > SLOT_BYTES) return 3;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 4;
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * bytes, c->slots + (size_t)r * SLOT_BYTES, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return 1;
    if (!barrier(c)) return 4;
    return 0;
}

ncclResult_t ncclGroupStart() { return 0; }   // (every call above is complete when it returns: nothing to group)
ncclResult_t ncclGroupEnd() { return 0; }
const char* ncclGetErrorString(ncclResult_t r) {
    return r == 0 ? "success" : (r == 1 ? "fake rccl: HIP error" : (r == 2 ? "fake rccl: shared memory" : (r == 4 ? "fake rccl: a rank never arrived" : "fake rccl: unsupported")));
}
}
