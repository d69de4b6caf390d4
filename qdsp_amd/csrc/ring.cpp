// ring.cpp -- the one exchange of the sharded path, as a C ABI on RCCL (include/qdsp_hip.h "ring").
//
// A long IQ stream is cut into time chunks over the GPUs of a node; a chunk's filter needs the LAST H input samples of the
// chunk before it in the stream (H = ntaps - 1 for the FIR, taps per phase for the resampler: src/dsp/filter.h:71,
// src/dsp/resampling.h:129 carry exactly those samples from call to call).  Every step each rank sends the tail of its chunk to
// its ring successor and receives its predecessor's:
//     ncclGroupStart(); ncclSend(tail, H samples, rank + 1); ncclRecv(halo, H samples, rank - 1); ncclGroupEnd();
// on the ring's own HIP stream -- 2 040 bytes at 256 taps, latency-bound, posted one step ahead so that it travels under the
// previous step's kernel.  No all-reduce, no all-gather.  qdsp_amd/sharding.py RingStream (bench.py --gpus N, the tests) and
// qdsp_amd/host/examples/graph_check.cpp `shard` drive this file; the reference has nothing of the kind (its only concurrency is
// one thread per block, src/dsp/block.h:83-85).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already carries one -- torch.distributed's -- gets that
// instance, and libqdsp_hip.so loads on a box without RCCL (single-GPU users never touch this file).
#include "../../include/qdsp_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

namespace {

struct UniqueId { char internal[QDSP_HIP_RING_ID_BYTES]; };     // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void* Comm;                                             // ncclComm_t

struct Rccl {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;      // (buf, count, ncclDataType_t, peer, comm, stream)
    int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;          // introspection only (qdsp_hip_ring_info): optional symbols
    int (*CommUserRank)(Comm, int*) = nullptr;
    int (*CommCuDevice)(Comm, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        // QDSP_RING_DISABLE_RCCL=1: behave as on a host without RCCL (the tests of the all-ranks-fall-back-together vote)
        const char* off = getenv("QDSP_RING_DISABLE_RCCL");
        if (off && off[0] == '1') return;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        Rccl r;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
        g_rccl = r;
    });
    return g_rccl;
}

constexpr unsigned kRingMagic = 0x52494e47u;   // "RING"
// Receive buffers in rotation.  complete(i) hands out recv[i % 4] (*d_halo) and recv[(i - 1) % 4] (*d_prev_halo: rank 0 of a
// block-cyclic stream reads what arrived a step earlier, qdsp_amd/sharding.py RingStream); with at most two posts outstanding the
// posts allowed after complete(i) are i + 1 and i + 2, which receive into recv[(i + 1) % 4] and recv[(i + 2) % 4] -- never into a
// buffer complete(i) handed out.  recv[k] is next written by post i + 4 (i + 3 for the prev pointer): that post makes the ring
// stream wait for `consumed`, recorded on the consumer stream at the start of complete(i + 2), i.e. behind every reader queued
// between complete(i) and complete(i + 2).
constexpr int kBufs = 4;
constexpr int kMaxInFlight = 2;
constexpr int kNcclInt8 = 0;                   // ncclInt8 / ncclChar

struct Ring {
    unsigned magic = kRingMagic;
    int device = 0, rank = 0, world = 1;
    size_t halo_bytes = 0;
    Comm comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;                // producer stream -> ring stream
    hipEvent_t done[kBufs] = {};               // ring stream -> consumer stream, one per receive buffer
    hipEvent_t consumed[kBufs] = {};           // consumer stream -> ring stream: consumed[j % 4] is recorded at the start of complete(j)
    hipEvent_t t0[kBufs] = {}, t1[kBufs] = {}; // timing mode only: around the send/recv group on the ring stream
    bool timed[kBufs] = {};
    void* recv[kBufs] = {};
    void* zeros = nullptr;
    long long posted = 0, completed = 0;
    bool timing = false;
    double us_sum = 0.0, us_max = 0.0;
    long long us_n = 0;
};

// timing mode: fold the finished exchange of slot k into the statistics (called when the slot is about to be reused, and by drain)
void ring_collect(Ring* r, int k, bool wait) {
    if (!r->timed[k]) return;
    if (wait) (void)hipEventSynchronize(r->t1[k]);
    else if (hipEventQuery(r->t1[k]) != hipSuccess) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r->t0[k], r->t1[k]) == hipSuccess) {
        const double us = (double)ms * 1e3;
        r->us_sum += us;
        if (us > r->us_max) r->us_max = us;
        r->us_n++;
    }
    r->timed[k] = false;
}

Ring* as_ring(void* h) {
    Ring* r = static_cast<Ring*>(h);
    return (r && r->magic == kRingMagic) ? r : nullptr;
}

#define RING_HIP(x)                                  \
    do {                                             \
        const hipError_t e_ = (x);                   \
        if (e_ != hipSuccess) return -(int)e_;       \
    } while (0)

int nccl_fail(const char* what, int rc) {
    const Rccl& n = rccl();
    fprintf(stderr, "qdsp_hip ring: %s failed: %s (%d)\n", what, n.GetErrorString ? n.GetErrorString(rc) : "?", rc);
    return QDSP_HIP_ERCCL;
}

void ring_free(Ring* r) {
    if (r->comm && rccl().ok) (void)rccl().CommDestroy(r->comm);
    for (int i = 0; i < kBufs; i++) {
        if (r->done[i]) (void)hipEventDestroy(r->done[i]);
        if (r->consumed[i]) (void)hipEventDestroy(r->consumed[i]);
        if (r->t0[i]) (void)hipEventDestroy(r->t0[i]);
        if (r->t1[i]) (void)hipEventDestroy(r->t1[i]);
        if (r->recv[i]) (void)hipFree(r->recv[i]);
    }
    if (r->zeros) (void)hipFree(r->zeros);
    if (r->ready) (void)hipEventDestroy(r->ready);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    r->magic = 0;
    delete r;
}

}  // namespace

extern "C" {

int qdsp_hip_ring_available(void) { return rccl().ok ? 1 : 0; }

int qdsp_hip_ring_unique_id(void* id) {
    if (!id) return QDSP_HIP_EINVAL;
    const Rccl& n = rccl();
    if (!n.ok) return QDSP_HIP_ERCCL;
    UniqueId u;
    const int rc = n.GetUniqueId(&u);
    if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, QDSP_HIP_RING_ID_BYTES);
    return 0;
}

int qdsp_hip_ring_create(void** ring, int device, int rank, int world, const void* id, int halo_bytes) {
    if (!ring || !id || world < 1 || rank < 0 || rank >= world || halo_bytes <= 0) return QDSP_HIP_EINVAL;
    *ring = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return QDSP_HIP_ENODEV;
    const Rccl& n = rccl();
    if (!n.ok) return QDSP_HIP_ERCCL;
    RING_HIP(hipSetDevice(device));
    Ring* r = new (std::nothrow) Ring;
    if (!r) return QDSP_HIP_ENOMEM;
    r->device = device;
    r->rank = rank;
    r->world = world;
    r->halo_bytes = (size_t)halo_bytes;
    hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ready, hipEventDisableTiming);
    for (int i = 0; i < kBufs && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->consumed[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&r->recv[i], r->halo_bytes);
        if (e == hipSuccess) e = hipMemset(r->recv[i], 0, r->halo_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(&r->zeros, r->halo_bytes);
    if (e == hipSuccess) e = hipMemset(r->zeros, 0, r->halo_bytes);
    if (e != hipSuccess) {
        ring_free(r);
        return -(int)e;
    }
    UniqueId u;
    memcpy(u.internal, id, QDSP_HIP_RING_ID_BYTES);
    const int rc = n.CommInitRank(&r->comm, world, u, rank);
    if (rc != 0) {
        r->comm = nullptr;
        ring_free(r);
        return nccl_fail("ncclCommInitRank", rc);
    }
    *ring = r;
    return 0;
}

int qdsp_hip_ring_post(void* ring, const void* d_tail, void* producer_stream) {
    Ring* r = as_ring(ring);
    if (!r || !d_tail) return QDSP_HIP_EINVAL;
    if (r->posted - r->completed >= kMaxInFlight) return QDSP_HIP_EINVAL;   // at most two in flight (see kBufs)
    const Rccl& n = rccl();
    RING_HIP(hipSetDevice(r->device));
    // the buffer this post receives into was last handed out by complete(posted - 4) / (posted - 3): wait for its readers
    if (r->posted >= 2 && r->completed >= r->posted - 1)
        RING_HIP(hipStreamWaitEvent(r->stream, r->consumed[(r->posted - 2) % kBufs], 0));
    // the tail is whatever the producer stream has written by now: the ring stream waits for that point, not for the host
    RING_HIP(hipEventRecord(r->ready, static_cast<hipStream_t>(producer_stream)));
    RING_HIP(hipStreamWaitEvent(r->stream, r->ready, 0));
    const int k = (int)(r->posted % kBufs);
    const int nxt = (r->rank + 1) % r->world, prv = (r->rank + r->world - 1) % r->world;
    if (r->timing) {
        ring_collect(r, k, false);
        if (!r->timed[k]) RING_HIP(hipEventRecord(r->t0[k], r->stream));
    }
    int rc = n.GroupStart();
    if (rc != 0) return nccl_fail("ncclGroupStart", rc);
    rc = n.Send(d_tail, r->halo_bytes, kNcclInt8, nxt, r->comm, r->stream);
    const int rc2 = n.Recv(r->recv[k], r->halo_bytes, kNcclInt8, prv, r->comm, r->stream);
    const int rc3 = n.GroupEnd();
    if (rc != 0) return nccl_fail("ncclSend", rc);
    if (rc2 != 0) return nccl_fail("ncclRecv", rc2);
    if (rc3 != 0) return nccl_fail("ncclGroupEnd", rc3);
    if (r->timing && !r->timed[k]) {
        RING_HIP(hipEventRecord(r->t1[k], r->stream));
        r->timed[k] = true;
    }
    RING_HIP(hipEventRecord(r->done[k], r->stream));
    r->posted++;
    return 0;
}

int qdsp_hip_ring_complete(void* ring, void* consumer_stream, const void** d_halo, const void** d_prev_halo) {
    Ring* r = as_ring(ring);
    if (!r || r->completed >= r->posted) return QDSP_HIP_EINVAL;
    RING_HIP(hipSetDevice(r->device));
    const int k = (int)(r->completed % kBufs);
    RING_HIP(hipEventRecord(r->consumed[k], static_cast<hipStream_t>(consumer_stream)));
    RING_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), r->done[k], 0));
    if (d_halo) *d_halo = r->recv[k];
    if (d_prev_halo) *d_prev_halo = r->completed > 0 ? r->recv[(k + kBufs - 1) % kBufs] : r->zeros;
    r->completed++;
    return 0;
}

int qdsp_hip_ring_drain(void* ring) {
    Ring* r = as_ring(ring);
    if (!r) return QDSP_HIP_EINVAL;
    RING_HIP(hipSetDevice(r->device));
    RING_HIP(hipStreamSynchronize(r->stream));
    r->completed = r->posted;
    for (int k = 0; k < kBufs; k++) ring_collect(r, k, true);
    return 0;
}

int qdsp_hip_ring_info(void* ring, int* comm_ranks, int* comm_rank, int* comm_device, int* rccl_version) {
    Ring* r = as_ring(ring);
    if (!r) return QDSP_HIP_EINVAL;
    const Rccl& n = rccl();
    // what the COMMUNICATOR says, not what create() was told: -1 where this RCCL build lacks the query
    int v = -1;
    if (comm_ranks) { v = -1; if (n.CommCount && n.CommCount(r->comm, &v) != 0) v = -1; *comm_ranks = v; }
    if (comm_rank) { v = -1; if (n.CommUserRank && n.CommUserRank(r->comm, &v) != 0) v = -1; *comm_rank = v; }
    if (comm_device) { v = -1; if (n.CommCuDevice && n.CommCuDevice(r->comm, &v) != 0) v = -1; *comm_device = v; }
    if (rccl_version) { v = -1; if (n.GetVersion && n.GetVersion(&v) != 0) v = -1; *rccl_version = v; }
    return 0;
}

int qdsp_hip_ring_set_timing(void* ring, int on) {
    Ring* r = as_ring(ring);
    if (!r) return QDSP_HIP_EINVAL;
    RING_HIP(hipSetDevice(r->device));
    if (on && !r->t0[0]) {
        for (int i = 0; i < kBufs; i++) {
            RING_HIP(hipEventCreate(&r->t0[i]));
            RING_HIP(hipEventCreate(&r->t1[i]));
        }
    }
    r->timing = on != 0;
    r->us_sum = r->us_max = 0.0;
    r->us_n = 0;
    return 0;
}

int qdsp_hip_ring_exchange_us(void* ring, double* mean_us, double* max_us, long long* exchanges) {
    Ring* r = as_ring(ring);
    if (!r) return QDSP_HIP_EINVAL;
    if (mean_us) *mean_us = r->us_n ? r->us_sum / (double)r->us_n : 0.0;
    if (max_us) *max_us = r->us_max;
    if (exchanges) *exchanges = r->us_n;
    return 0;
}

void qdsp_hip_ring_destroy(void* ring) {
    Ring* r = as_ring(ring);
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    ring_free(r);
}

}  // extern "C"
