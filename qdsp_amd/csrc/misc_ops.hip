// misc_ops.hip -- the element-wise two-input blocks of src/dsp/math.h (Add / Substract / Multiply), the synthetic IQ source of the
// bench, completion events and the timing / introspection helpers of the harness.  Split out of qdsp_hip.hip in round 3.
#include "engine.hip.h"

namespace qh {

// ---- element-wise two-input operator (src/dsp/math.h) ----------------------------------------
constexpr uint32_t kMathMagic = 0x514d4154;  // "QMAT"
struct Math {
    uint32_t magic = kMathMagic;
    int device = 0, op = 0, ch = 1;
    hipStream_t stream = nullptr;
    float *d_a = nullptr, *d_b = nullptr, *d_out = nullptr;
    int max_block = 0;
};
Math* as_math(void* h) {
    Math* m = static_cast<Math*>(h);
    return (m && m->magic == kMathMagic) ? m : nullptr;
}
int math_ensure(Math* m, int count) {
    if (count <= m->max_block) return 0;
    HIPCHK(hipSetDevice(m->device));
    for (float** p : {&m->d_a, &m->d_b, &m->d_out}) {
        if (*p) HIPCHK(hipFree(*p));
        *p = nullptr;
        HIPCHK(hipMalloc(p, (size_t)count * m->ch * sizeof(float)));
    }
    m->max_block = count;
    return 0;
}
int math_launch(Math* m, const void* d_a, const void* d_b, int64_t count, void* d_out, hipStream_t s) {
    if (count <= 0) return 0;
    const uintptr_t al = (uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_out;
    if (al & 15) return QDSP_HIP_EINVAL;   // device buffers come from hipMalloc / stream<T>: 16-byte aligned
    qk::EwArgs a;
    a.a = static_cast<const float*>(d_a);
    a.b = static_cast<const float*>(d_b);
    a.out = static_cast<float*>(d_out);
    a.n = count * m->ch;
    long long n4 = a.n >> 2;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 256 * 1024) grid = 256 * 1024;   // (one float4 per lane: more memory-level parallelism than a grid-stride loop)
    if (grid < 1) grid = 1;
    const bool cx = m->ch == 2;
#define QK_EW(op, c) hipLaunchKernelGGL((qk::ew_kernel<op, c>), dim3(grid), dim3(256), 0, s, a)
    switch (m->op) {
        case 0: QK_EW(0, false); break;
        case 1: QK_EW(1, false); break;
        default: if (cx) QK_EW(2, true); else QK_EW(2, false); break;
    }
#undef QK_EW
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace qh

using namespace qh;

extern "C" {

// ---- element-wise math blocks --------------------------------------------------------------------
void qdsp_hip_math_destroy(void* h);
int qdsp_hip_math_create(void** h, int device, int op, int complex_data, int max_block) {
    if (!h || op < 0 || op > 2) return QDSP_HIP_EINVAL;
    *h = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QDSP_HIP_ENODEV;
    if (device < 0 || device >= ndev) return QDSP_HIP_EINVAL;
    Math* m = new (std::nothrow) Math();
    if (!m) return QDSP_HIP_ENOMEM;
    m->device = device;
    m->op = op;
    m->ch = complex_data ? 2 : 1;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        delete m;
        return QDSP_HIP_ENOMEM;
    }
    if (max_block > 0) { int rc = math_ensure(m, max_block); if (rc) { qdsp_hip_math_destroy(m); return rc; } }
    *h = m;
    return 0;
}
int qdsp_hip_math_process_dev(void* h, const void* d_a, const void* d_b, int64_t count, void* d_out, void* s) {
    Math* m = as_math(h);
    if (!m || count < 0 || (count > 0 && (!d_a || !d_b || !d_out))) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(m->device));
    return math_launch(m, d_a, d_b, count, d_out, static_cast<hipStream_t>(s));
}
int qdsp_hip_math_process_ex(void* h, const void* a, int a_dev, const void* b, int b_dev, int count, void* out, int out_dev) {
    Math* m = as_math(h);
    if (!m || count < 0 || (count > 0 && (!a || !b || !out))) return QDSP_HIP_EINVAL;
    if (count == 0) return 0;
    if (!a_dev || !b_dev || !out_dev) { int rc = math_ensure(m, count); if (rc) return rc; }
    HIPCHK(hipSetDevice(m->device));
    const size_t bytes = (size_t)count * m->ch * sizeof(float);
    const void *sa = a, *sb = b;
    if (!a_dev) { HIPCHK(hipMemcpyAsync(m->d_a, a, bytes, hipMemcpyHostToDevice, m->stream)); sa = m->d_a; }
    if (!b_dev) { HIPCHK(hipMemcpyAsync(m->d_b, b, bytes, hipMemcpyHostToDevice, m->stream)); sb = m->d_b; }
    void* dst = out_dev ? out : m->d_out;
    int rc = math_launch(m, sa, sb, count, dst, m->stream);
    if (rc) return rc;
    if (!out_dev) HIPCHK(hipMemcpyAsync(out, m->d_out, bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(wait_stream(m->stream));
    return 0;
}
int qdsp_hip_math_process(void* h, const void* a, const void* b, int count, void* out) {
    return qdsp_hip_math_process_ex(h, a, 0, b, 0, count, out, 0);
}
void qdsp_hip_math_destroy(void* h) {
    Math* m = as_math(h);
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    for (float* p : {m->d_a, m->d_b, m->d_out})
        if (p) (void)hipFree(p);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    m->magic = 0;
    delete m;
}

// ---- harness ----------------------------------------------------------------------------------
int qdsp_hip_synth_iq_dev(int device, void* d_out, int64_t first_sample, int64_t count, uint32_t seed, void* stream) {
    if (count < 0 || (count > 0 && !d_out)) return QDSP_HIP_EINVAL;
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(device));
    constexpr int NT = 256;
    const uint32_t key = qk::mix32(seed * 0x9e3779b9U + 0x85ebca6bU);
    long long grid = (2 * count / 4 + NT - 1) / NT + 1;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL((qk::synth_iq_kernel<NT>), dim3((unsigned)grid), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(d_out), (long long)first_sample, (long long)count, key);
    HIPCHK(hipGetLastError());
    return 0;
}

// Completion events for QDSP_HIP_LINK_HOST_DEFERRED (the consumer of a host buffer waits, not the producer).
int qdsp_hip_event_create(int device, void** ev) {
    if (!ev) return QDSP_HIP_EINVAL;
    *ev = nullptr;
    HIPCHK(hipSetDevice(device));
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return 0;
}
int qdsp_hip_event_destroy(void* ev) {
    if (ev) HIPCHK(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return 0;
}
int qdsp_hip_event_wait(void* ev) {
    if (!ev) return QDSP_HIP_EINVAL;
    hipEvent_t e = static_cast<hipEvent_t>(ev);
    static const int spin_us = 200;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        do {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) return -(int)q;
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us));
    }
    HIPCHK(hipEventSynchronize(e));
    return 0;
}
int qdsp_hip_set_done_event(void* h, void* ev) {
    Engine* e = any_engine(h);
    if (!e) return QDSP_HIP_EINVAL;
    e->done_ev = static_cast<hipEvent_t>(ev);
    return 0;
}

int qdsp_hip_last_kernel(void* h, char* name, int name_len, int* grid, int* block, int* lds) {
    if (Chan* c = as_chan(h)) {
        if (name && name_len > 0) { strncpy(name, c->last.name, name_len - 1); name[name_len - 1] = 0; }
        if (grid) *grid = c->last.grid;
        if (block) *block = c->last.block;
        if (lds) *lds = c->last.lds;
        return 0;
    }
    Engine* e = any_engine(h);
    if (!e) return QDSP_HIP_EINVAL;
    if (name && name_len > 0) { strncpy(name, e->last.name, name_len - 1); name[name_len - 1] = 0; }
    if (grid) *grid = e->last.grid;
    if (block) *block = e->last.block;
    if (lds) *lds = e->last.lds;
    return 0;
}

int qdsp_hip_time_process_dev(void* h, const void* d_in, int64_t count, void* d_out, void* stream, int iters, float* ms) {
    Engine* e = any_engine(h);
    return e ? time_process(e, d_in, count, d_out, stream, iters, ms) : QDSP_HIP_EINVAL;
}

}  // extern "C"
