// TEST-ONLY: the dozen HIP runtime entry points and the RCCL entry points qdsp_amd/csrc/ring.cpp uses, as synchronous host code, so
// that the ring's bookkeeping (four receive buffers in rotation, at most two posts outstanding, prev / zeros pointers, timing slots)
// runs under -fsanitize=address,undefined without a GPU (tests/test_sanitizers_cpu.py).  Built into two shared objects:
// libfakehip.so (-DFAKE_HIP, linked instead of libamdhip64) and librccl.so.1 (-DFAKE_RCCL: what ring.cpp dlopens; found first
// through LD_LIBRARY_PATH).  A "stream" executes everything at once, so an event is complete as soon as it is recorded; a
// one-rank communicator's ncclSend / ncclRecv pair is a memcpy at ncclGroupEnd.
#include <stdlib.h>
#include <string.h>

extern "C" {

#ifdef FAKE_HIP
int hipGetDeviceCount(int* n) { *n = 1; return 0; }
int hipSetDevice(int) { return 0; }
int hipStreamCreateWithFlags(void** s, unsigned) { *s = malloc(16); return *s ? 0 : 2; }
int hipStreamDestroy(void* s) { free(s); return 0; }
int hipStreamSynchronize(void*) { return 0; }
int hipStreamWaitEvent(void*, void* e, unsigned) { return e ? 0 : 1; }
int hipEventCreateWithFlags(void** e, unsigned) { *e = calloc(1, 16); return *e ? 0 : 2; }
int hipEventCreate(void** e) { return hipEventCreateWithFlags(e, 0); }
int hipEventDestroy(void* e) { free(e); return 0; }
int hipEventRecord(void* e, void*) { if (!e) return 1; *static_cast<int*>(e) += 1; return 0; }
int hipEventSynchronize(void* e) { return e ? 0 : 1; }
int hipEventQuery(void* e) { return e ? 0 : 1; }
int hipEventElapsedTime(float* ms, void* a, void* b) { if (!a || !b) return 1; *ms = 0.002f; return 0; }
int hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
int hipFree(void* p) { free(p); return 0; }
int hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
#endif

#ifdef FAKE_RCCL
struct UniqueId { char internal[128]; };
struct Comm { int world, rank; };
static const void* g_send = nullptr;
static void* g_recv = nullptr;
static size_t g_send_n = 0, g_recv_n = 0;
static int g_depth = 0;
int ncclGetUniqueId(UniqueId* id) { memset(id->internal, 0x5a, sizeof id->internal); return 0; }
int ncclCommInitRank(Comm** c, int world, UniqueId, int rank) {
    if (world != 1 || rank != 0) return 5;      // the fake carries a one-rank ring only
    *c = static_cast<Comm*>(malloc(sizeof(Comm)));
    (*c)->world = world;
    (*c)->rank = rank;
    return 0;
}
int ncclCommDestroy(Comm* c) { free(c); return 0; }
int ncclGroupStart() { g_depth++; return 0; }
int ncclSend(const void* buf, size_t count, int, int peer, Comm* c, void*) { if (peer != 0 || !c) return 4; g_send = buf; g_send_n = count; return 0; }
int ncclRecv(void* buf, size_t count, int, int peer, Comm* c, void*) { if (peer != 0 || !c) return 4; g_recv = buf; g_recv_n = count; return 0; }
int ncclGroupEnd() {
    if (--g_depth == 0 && g_send && g_recv) {
        if (g_send_n != g_recv_n) return 4;
        memcpy(g_recv, g_send, g_send_n);
        g_send = nullptr;
        g_recv = nullptr;
    }
    return 0;
}
const char* ncclGetErrorString(int) { return "fake rccl error"; }
int ncclCommCount(Comm* c, int* n) { *n = c->world; return 0; }
int ncclCommUserRank(Comm* c, int* r) { *r = c->rank; return 0; }
int ncclCommCuDevice(Comm*, int* d) { *d = 0; return 0; }
int ncclGetVersion(int* v) { *v = 22203; return 0; }
#endif

}  // extern "C"
