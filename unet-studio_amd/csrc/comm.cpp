// RCCL under the C ABI (include/unet_hip.h: unet_comm_*).  Replaces the reference's replica synchronisation -- the per-parameter
// reduce-to-root `grad.to(device0); add_` of UNet3dImpl::add_gradient_from (unet.cpp:224-244, train.cpp:756-757) and the per-step
// weight broadcast of copy_from (unet.cpp:195-222, train.cpp:573-579) -- by collectives over xGMI on the flat buffers.
//
// librccl is NOT a link-time dependency: the first unet_comm_* call binds the copy that is already in the process (PyTorch-ROCm's
// bundled librccl.so, which libtorch_hip pulls in) or loads one by name, so CPU-only hosts and the parity tests never touch it and
// a process never ends up with two RCCL instances.  Collectives run on a stream owned by the communicator, ordered after the
// caller's stream by an event: the all-reduce of a finished gradient bucket overlaps the rest of the backward
// (unet_backward_part), and unet_comm_join orders the caller's stream after everything the communicator has been given.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/unet_hip.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;            // the copy already in the process (torch's)
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return;
        auto sym = [&](const char* s) { return dlsym(r.h, s); };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GroupStart ||
        !r.GroupEnd)
        throw std::runtime_error("unet_comm: librccl.so is not available in this process (import torch / link libtorch_hip first, or put "
                                 "/opt/rocm/lib on the library path)");
    return r;
}

void nccl_ok(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) {
        Rccl& r = rccl();
        throw std::runtime_error(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
    }
}
void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        hip_ok(hipGetDevice(&prev), "hipGetDevice");
        if (prev != dev) hip_ok(hipSetDevice(dev), "hipSetDevice"); else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct unet_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;          // collectives run here
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    ~unet_comm() {
        if (comm) (void)rccl().CommDestroy(comm);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {
void make_streams(unet_comm* c) {
    DevGuard g(c->device);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hip_ok(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi), "hipStreamCreateWithPriority");   // collectives first: they gate the update
    hip_ok(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming), "hipEventCreate");
    hip_ok(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming), "hipEventCreate");
}
void fork_from(unet_comm* c, void* stream) {
    hip_ok(hipEventRecord(c->ev_in, (hipStream_t)stream), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(c->stream, c->ev_in, 0), "hipStreamWaitEvent");
}
}  // namespace

#define COMM_TRY(...)                                                        \
    try { __VA_ARGS__; return 0; } catch (const std::exception& e) { unet_set_error(e.what()); return 1; }

extern "C" {

int unet_comm_unique_id(void* id_bytes) {
    COMM_TRY({
        if (!id_bytes) throw std::runtime_error("unet_comm_unique_id: null argument");
        static_assert(sizeof(ncclUniqueId) == UNET_COMM_ID_BYTES, "id size");
        nccl_ok(rccl().GetUniqueId((ncclUniqueId*)id_bytes), "ncclGetUniqueId");
    })
}

int unet_comm_create(int rank, int world, const void* id_bytes, int device, unet_comm** out) {
    COMM_TRY({
        if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) throw std::runtime_error("unet_comm_create: bad argument");
        std::unique_ptr<unet_comm> c(new unet_comm());
        c->rank = rank; c->world = world; c->device = device;
        DevGuard g(device);
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        nccl_ok(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
        make_streams(c.get());
        *out = c.release();
    })
}

int unet_comm_create_all(int n, const int* devices, unet_comm** out) {
    COMM_TRY({
        if (n < 1 || !devices || !out) throw std::runtime_error("unet_comm_create_all: bad argument");
        std::vector<ncclComm_t> cs(n, nullptr);
        nccl_ok(rccl().CommInitAll(cs.data(), n, devices), "ncclCommInitAll");
        // every communicator gets an owner before anything else can throw: a failure half way (stream / event creation on device i)
        // destroys ALL n RCCL communicators and whatever streams exist, and leaves out[] untouched
        std::vector<std::unique_ptr<unet_comm>> owned;
        for (int i = 0; i < n; ++i) {
            owned.emplace_back(new unet_comm());
            owned.back()->comm = cs[i]; owned.back()->rank = i; owned.back()->world = n; owned.back()->device = devices[i];
        }
        for (int i = 0; i < n; ++i) make_streams(owned[i].get());
        for (int i = 0; i < n; ++i) out[i] = owned[i].release();
    })
}

int unet_comm_destroy(unet_comm* c) {
    COMM_TRY({
        if (c) { DevGuard g(c->device); delete c; }
    })
}

int unet_comm_rank(const unet_comm* c, int* rank, int* world) {
    if (!c) return 1;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return 0;
}

int unet_allreduce_grads(unet_comm* c, float* flat, int64_t elem_lo, int64_t elem_hi, void* stream) {
    COMM_TRY({
        if (!c || !flat || elem_lo < 0 || elem_hi < elem_lo) throw std::runtime_error("unet_allreduce_grads: bad argument");
        if (elem_hi == elem_lo) return 0;
        DevGuard g(c->device);
        fork_from(c, stream);    // flat[lo:hi] is final on the caller's stream at this point
        nccl_ok(rccl().AllReduce(flat + elem_lo, flat + elem_lo, (size_t)(elem_hi - elem_lo), ncclFloat32, ncclSum, c->comm, c->stream),
                "ncclAllReduce");
    })
}

int unet_allreduce_grads_all(unet_comm* const* comms, int n, float* const* flats, int64_t elem_lo, int64_t elem_hi, void* const* streams) {
    COMM_TRY({
        if (!comms || !flats || !streams || n < 1 || elem_lo < 0 || elem_hi < elem_lo) throw std::runtime_error("unet_allreduce_grads_all: bad argument");
        for (int i = 0; i < n; ++i)
            if (!comms[i] || !flats[i]) throw std::runtime_error("unet_allreduce_grads_all: null communicator or buffer at index " + std::to_string(i));
        if (elem_hi == elem_lo) return 0;
        for (int i = 0; i < n; ++i) { DevGuard g(comms[i]->device); fork_from(comms[i], streams[i]); }
        nccl_ok(rccl().GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; ++i)
            nccl_ok(rccl().AllReduce(flats[i] + elem_lo, flats[i] + elem_lo, (size_t)(elem_hi - elem_lo), ncclFloat32, ncclSum, comms[i]->comm,
                                     comms[i]->stream), "ncclAllReduce");
        nccl_ok(rccl().GroupEnd(), "ncclGroupEnd");
    })
}

int unet_comm_broadcast(unet_comm* c, float* buf, int64_t n, int root, void* stream) {
    COMM_TRY({
        if (!c || !buf || n < 0 || root < 0 || root >= c->world) throw std::runtime_error("unet_comm_broadcast: bad argument");
        if (n == 0) return 0;
        DevGuard g(c->device);
        fork_from(c, stream);
        nccl_ok(rccl().Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, c->stream), "ncclBroadcast");
    })
}

int unet_comm_join(unet_comm* c, void* stream) {
    COMM_TRY({
        if (!c) throw std::runtime_error("unet_comm_join: null communicator");
        DevGuard g(c->device);
        hip_ok(hipEventRecord(c->ev_out, c->stream), "hipEventRecord");
        hip_ok(hipStreamWaitEvent((hipStream_t)stream, c->ev_out, 0), "hipStreamWaitEvent");
    })
}

}  // extern "C"
