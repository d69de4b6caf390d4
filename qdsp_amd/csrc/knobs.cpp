// knobs.cpp -- snapshot of the QDSP_HIP_* environment variables (see knobs.h).  Host code only.
#include "knobs.h"

#include "../../include/qdsp_hip.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

namespace qk {
namespace {

const char* const kNames[K_COUNT] = {
#define X(n) "QDSP_HIP_" #n,
    QDSP_HIP_KNOBS(X)
#undef X
};

std::atomic<const Knobs*> g_knobs{nullptr};
std::mutex g_mtx;

const Knobs* build() {
    Knobs* k = new Knobs;
    memset(k, 0, sizeof(*k));
    for (int i = 0; i < K_COUNT; i++) {
        const char* s = getenv(kNames[i]);
        if (s && *s) {
            k->val[i] = atoi(s);
            k->set[i] = 1;
        }
    }
    if (const char* st = getenv("QDSP_HIP_FFT_STAMPS")) k->fft_stamps = strtoull(st, nullptr, 0);
    return k;
}

}  // namespace

const Knobs* knobs_snapshot() {
    const Knobs* k = g_knobs.load(std::memory_order_acquire);
    if (k) return k;
    std::lock_guard<std::mutex> lk(g_mtx);
    k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        k = build();
        g_knobs.store(k, std::memory_order_release);
    }
    return k;
}

void knobs_reload() {
    std::lock_guard<std::mutex> lk(g_mtx);
    g_knobs.store(build(), std::memory_order_release);   // (the previous table stays allocated: a reader may still hold it)
}

}  // namespace qk

extern "C" int qdsp_hip_reload_env(void) {
    qk::knobs_reload();
    return 0;
}
