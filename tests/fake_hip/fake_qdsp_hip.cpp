// fake_qdsp_hip.cpp -- TEST-ONLY stand-in for libqdsp_hip.so: the same C ABI (include/qdsp_hip.h), no GPU, no arithmetic.
//
// Purpose: the block-graph mirror under qdsp_amd/host/dsp is ~2 000 lines of lock- and
// thread-heavy C++ (stream hand-offs, device-resident links, Splitter banks, live retunes).  GPU AddressSanitizer is not
// available on the test pool, so its HOST logic is run under -fsanitize=thread and -fsanitize=address,undefined against this
// library: "device" memory is malloc, every operator copies (or decimates) its input into its output on the calling thread,
// events are plain objects.  Results are meaningless as signal processing; what counts is that every graph runs to completion
// and the sanitizers stay silent.  Never shipped, never loaded by the product or by the parity tests
// (tests/test_sanitizers_cpu.py builds it into tests/fake_hip/build/ with the sanitizer flags and links graph_check against it).
#include "../../include/qdsp_hip.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <new>
#include <thread>

namespace {

constexpr unsigned kMagic = 0x46414b45u;   // "FAKE"

struct Op {
    unsigned magic = kMagic;
    int elem = 8;                // bytes per sample: 8 = float pair, 4 = float
    int interp = 1, decim = 1;
    int nchan = 1;
    std::atomic<long long> calls{0};
    std::atomic<void*> done{nullptr};
};

Op* op(void* h) {
    Op* o = static_cast<Op*>(h);
    return (o && o->magic == kMagic) ? o : nullptr;
}

int make(void** h, int elem, int interp, int decim, int nchan = 1) {
    if (!h || interp < 1 || decim < 1 || nchan < 1) return QDSP_HIP_EINVAL;
    Op* o = new (std::nothrow) Op;
    if (!o) return QDSP_HIP_ENOMEM;
    o->elem = elem;
    o->interp = interp;
    o->decim = decim;
    o->nchan = nchan;
    *h = o;
    return 0;
}

void unmake(void* h) {
    Op* o = op(h);
    if (!o) return;
    o->magic = 0;
    delete o;
}

long long out_count(const Op* o, long long count) { return count * o->interp / o->decim; }

// FAKE_HIP_JITTER=1: calls take an irregular amount of time (what kernels and copies do), so that the graph's threads meet in more than
// one order from run to run
void jitter(const Op* o) {
    static const bool on = [] { const char* e = getenv("FAKE_HIP_JITTER"); return e && e[0] == '1'; }();
    if (!on) return;
    const long long k = o->calls.load(std::memory_order_relaxed) * 2654435761LL + (long long)(size_t)o;
    if ((k >> 4) % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(20 + (k >> 8) % 180));
    else if ((k >> 4) % 3 == 0) std::this_thread::yield();
}

// out[n] = in[(n * decim) / interp]: touches every byte of `out` the real kernel would write and reads inside `in`
long long resample_copy(Op* o, const void* in, long long count, void* out) {
    if (count < 0 || (count > 0 && (!in || !out))) return QDSP_HIP_EINVAL;
    const long long no = out_count(o, count);
    jitter(o);
    const char* s = static_cast<const char*>(in);
    char* d = static_cast<char*>(out);
    if (o->interp == 1 && o->decim == 1) {
        if (count) memcpy(d, s, (size_t)count * o->elem);
    } else {
        for (long long n = 0; n < no; n++) memcpy(d + n * o->elem, s + (n * o->decim / o->interp) * o->elem, (size_t)o->elem);
    }
    o->calls++;
    return no;
}

}  // namespace

extern "C" {

int qdsp_hip_abi_version(void) { return 0x7fff; }
const char* qdsp_hip_error_string(int code) { return code == 0 ? "ok (fake)" : "error (fake libqdsp_hip)"; }
int qdsp_hip_device_count(int* count) { if (count) *count = 1; return 0; }
int qdsp_hip_reload_env(void) { return 0; }

int qdsp_hip_event_create(int, void** ev) {
    if (!ev) return QDSP_HIP_EINVAL;
    *ev = malloc(8);
    return *ev ? 0 : QDSP_HIP_ENOMEM;
}
int qdsp_hip_event_destroy(void* ev) { free(ev); return 0; }
int qdsp_hip_event_wait(void* ev) { return ev ? 0 : QDSP_HIP_EINVAL; }
int qdsp_hip_set_done_event(void* handle, void* ev) {
    Op* o = op(handle);
    if (!o) return QDSP_HIP_EINVAL;
    o->done.store(ev);
    return 0;
}

int qdsp_hip_host_alloc(void** p, size_t bytes) { if (!p) return QDSP_HIP_EINVAL; *p = calloc(1, bytes ? bytes : 1); return *p ? 0 : QDSP_HIP_ENOMEM; }
int qdsp_hip_host_free(void* p) { free(p); return 0; }
int qdsp_hip_host_register(void*, size_t) { return 0; }
int qdsp_hip_host_unregister(void*) { return 0; }
int qdsp_hip_dev_alloc(int, void** p, size_t bytes) { return qdsp_hip_host_alloc(p, bytes); }
int qdsp_hip_dev_free(int, void* p) { free(p); return 0; }
int qdsp_hip_memcpy_h2d(int, void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int qdsp_hip_memcpy_d2h(int, void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int qdsp_hip_memcpy_d2d(int, void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int qdsp_hip_memcpy_d2d_link(int, void* d, const void* s, size_t n, int, int) { if (n) memcpy(d, s, n); return 0; }
int qdsp_hip_memcpy_d2h_link(int, void* d, const void* s, size_t n, int) { if (n) memcpy(d, s, n); return 0; }
int qdsp_hip_device_sync(int) { return 0; }

// ---- FIR ----
int qdsp_hip_fir_cf32_create(void** h, int, const float* taps, int ntaps, int) { return (taps && ntaps > 0) ? make(h, 8, 1, 1) : QDSP_HIP_EINVAL; }
int qdsp_hip_fir_f32_create(void** h, int, const float* taps, int ntaps, int) { return (taps && ntaps > 0) ? make(h, 4, 1, 1) : QDSP_HIP_EINVAL; }
int qdsp_hip_fir_cf32_process_ex(void* h, const void* in, int, int count, void* out, int) { Op* o = op(h); if (!o) return QDSP_HIP_EINVAL; const long long r = resample_copy(o, in, count, out); return r < 0 ? (int)r : 0; }
int qdsp_hip_fir_f32_process_ex(void* h, const void* in, int a, int count, void* out, int b) { return qdsp_hip_fir_cf32_process_ex(h, in, a, count, out, b); }
int qdsp_hip_fir_cf32_process_dev(void* h, const void* in, int64_t count, void* out, void*) { Op* o = op(h); if (!o) return QDSP_HIP_EINVAL; const long long r = resample_copy(o, in, count, out); return r < 0 ? (int)r : 0; }
int qdsp_hip_fir_cf32_set_taps(void* h, const float* t, int n) { return (op(h) && t && n > 0) ? 0 : QDSP_HIP_EINVAL; }
int qdsp_hip_fir_f32_set_taps(void* h, const float* t, int n) { return (op(h) && t && n > 0) ? 0 : QDSP_HIP_EINVAL; }
int qdsp_hip_fir_cf32_set_history_dev(void* h, const void* d, void*) { return (op(h) && d) ? 0 : QDSP_HIP_EINVAL; }
void qdsp_hip_fir_cf32_destroy(void* h) { unmake(h); }
void qdsp_hip_fir_f32_destroy(void* h) { unmake(h); }

// ---- PolyphaseResampler ----
int qdsp_hip_decim_cf32_create(void** h, int, const float* taps, int ntaps, int interp, int decim, int) { return (taps && ntaps > 0) ? make(h, 8, interp, decim) : QDSP_HIP_EINVAL; }
int qdsp_hip_decim_f32_create(void** h, int, const float* taps, int ntaps, int interp, int decim, int) { return (taps && ntaps > 0) ? make(h, 4, interp, decim) : QDSP_HIP_EINVAL; }
int qdsp_hip_decim_cf32_process_ex(void* h, const void* in, int, int count, void* out, int) { Op* o = op(h); return o ? (int)resample_copy(o, in, count, out) : QDSP_HIP_EINVAL; }
int qdsp_hip_decim_f32_process_ex(void* h, const void* in, int a, int count, void* out, int b) { return qdsp_hip_decim_cf32_process_ex(h, in, a, count, out, b); }
int qdsp_hip_decim_cf32_configure(void* h, const float* t, int n, int interp, int decim) {
    Op* o = op(h);
    if (!o || !t || n <= 0 || interp < 1 || decim < 1) return QDSP_HIP_EINVAL;
    o->interp = interp;
    o->decim = decim;
    return 0;
}
int qdsp_hip_decim_f32_configure(void* h, const float* t, int n, int interp, int decim) { return qdsp_hip_decim_cf32_configure(h, t, n, interp, decim); }
void qdsp_hip_decim_cf32_destroy(void* h) { unmake(h); }
void qdsp_hip_decim_f32_destroy(void* h) { unmake(h); }

// ---- FrequencyXlator / SineSource ----
int qdsp_hip_xlate_cf32_create(void** h, int, float, float, int) { return make(h, 8, 1, 1); }
int qdsp_hip_xlate_cf32_process_ex(void* h, const void* in, int, int count, void* out, int) { Op* o = op(h); if (!o) return QDSP_HIP_EINVAL; const long long r = resample_copy(o, in, count, out); return r < 0 ? (int)r : 0; }
int qdsp_hip_xlate_cf32_set_phase_inc(void* h, float, float) { return op(h) ? 0 : QDSP_HIP_EINVAL; }
void qdsp_hip_xlate_cf32_destroy(void* h) { unmake(h); }
int qdsp_hip_sine_cf32_create(void** h, int, float, float, int) { return make(h, 8, 1, 1); }
int qdsp_hip_sine_cf32_generate(void* h, int count, void* out, int) {
    Op* o = op(h);
    if (!o || count < 0 || (count && !out)) return QDSP_HIP_EINVAL;
    if (count) memset(out, 0, (size_t)count * 8);
    o->calls++;
    return 0;
}
int qdsp_hip_sine_cf32_set_phase_inc(void* h, float, float) { return op(h) ? 0 : QDSP_HIP_EINVAL; }
void qdsp_hip_sine_cf32_destroy(void* h) { unmake(h); }

// ---- VFO (fused) ----
int qdsp_hip_xlate_fir_decim_cf32_create(void** h, int, const float* taps, int ntaps, int interp, int decim, float, float, int) { return (taps && ntaps > 0) ? make(h, 8, interp, decim) : QDSP_HIP_EINVAL; }
int qdsp_hip_xlate_fir_decim_cf32_process_ex(void* h, const void* in, int, int count, void* out, int) { Op* o = op(h); return o ? (int)resample_copy(o, in, count, out) : QDSP_HIP_EINVAL; }
int qdsp_hip_xlate_fir_decim_cf32_configure(void* h, const float* t, int n, int interp, int decim) { return qdsp_hip_decim_cf32_configure(h, t, n, interp, decim); }
int qdsp_hip_xlate_fir_decim_cf32_set_phase_inc(void* h, float, float) { return op(h) ? 0 : QDSP_HIP_EINVAL; }
void qdsp_hip_xlate_fir_decim_cf32_destroy(void* h) { unmake(h); }

// ---- channel bank (Splitter -> N x VFO) ----
int qdsp_hip_chan_cf32_create(void** h, int, const float* taps, int ntaps, int interp, int decim, int nchan, const float* re, const float* im, int) {
    return (taps && ntaps > 0 && re && im) ? make(h, 8, interp, decim, nchan) : QDSP_HIP_EINVAL;
}
int64_t qdsp_hip_chan_cf32_process_links(void* h, const void* in, int, int count, void* const* outs, const int* out_links, void*) {
    Op* o = op(h);
    if (!o || !outs || !out_links) return QDSP_HIP_EINVAL;
    long long no = 0;
    for (int c = 0; c < o->nchan; c++) {
        no = resample_copy(o, in, count, outs[c]);
        if (no < 0) return no;
    }
    return no;
}
int qdsp_hip_chan_cf32_move_channel_state(void* h, int chan, void* vfo, int) {
    Op* o = op(h);
    return (o && op(vfo) && chan >= 0 && chan < o->nchan) ? 0 : QDSP_HIP_EINVAL;
}
int qdsp_hip_chan_cf32_set_phase_inc(void* h, int chan, float, float) {
    Op* o = op(h);
    return (o && chan >= 0 && chan < o->nchan) ? 0 : QDSP_HIP_EINVAL;
}
void qdsp_hip_chan_cf32_destroy(void* h) { unmake(h); }

// ---- math blocks ----
int qdsp_hip_math_create(void** h, int, int op_, int complex_data, int) { return (op_ >= 0 && op_ <= 2) ? make(h, complex_data ? 8 : 4, 1, 1) : QDSP_HIP_EINVAL; }
int qdsp_hip_math_process_ex(void* h, const void* a, int, const void* b, int, int count, void* out, int) {
    Op* o = op(h);
    if (!o || count < 0 || (count && (!a || !b || !out))) return QDSP_HIP_EINVAL;
    // read BOTH inputs in full (a graph that hands over a stale or freed second input is what the sanitizers should see)
    const char* pa = static_cast<const char*>(a);
    const char* pb = static_cast<const char*>(b);
    char* po = static_cast<char*>(out);
    for (size_t i = 0, n = (size_t)count * o->elem; i < n; i++) po[i] = (char)(pa[i] ^ pb[i]);
    o->calls++;
    return 0;
}
void qdsp_hip_math_destroy(void* h) { unmake(h); }

// ---- ring: not available in the fake (graph_check `shard` is a GPU test) ----
int qdsp_hip_ring_available(void) { return 0; }
int qdsp_hip_ring_unique_id(void*) { return QDSP_HIP_ERCCL; }
int qdsp_hip_ring_create(void**, int, int, int, const void*, int) { return QDSP_HIP_ERCCL; }
int qdsp_hip_ring_post(void*, const void*, void*) { return QDSP_HIP_ERCCL; }
int qdsp_hip_ring_complete(void*, void*, const void**, const void**) { return QDSP_HIP_ERCCL; }
int qdsp_hip_ring_drain(void*) { return QDSP_HIP_ERCCL; }
void qdsp_hip_ring_destroy(void*) {}

}  // extern "C"
