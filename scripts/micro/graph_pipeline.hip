// graph_pipeline.hip -- what the GPU side of a device-resident block graph can deliver per block, issued from ONE host thread (no
// thread hand-offs): SineSource -> fused VFO (401 taps, decimate by 50) [-> second stage], `n` samples per block, in three orderings:
//   one     every launch on one in-order stream (rounds 1-3's pipelined links)
//   lanes   block k's launches on stream k % 2; a handle's consecutive calls ordered by an event (two-lane links)
//   stages  each operator on its OWN stream; data-ready and buffer-free events on the links (two buffers per link)
// build: hipcc -O2 -o graph_pipeline graph_pipeline.hip -I../../include -L../../qdsp_amd/csrc -lqdsp_hip -Wl,-rpath,$PWD/../../qdsp_amd/csrc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "qdsp_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, nblk = argc > 2 ? atoi(argv[2]) : 400;
    const int M = 50, NT = 401;
    std::vector<float> taps(NT);
    double sum = 0;
    for (int i = 0; i < NT; i++) { const double x = (i - 200) * 0.4 / M * 2; taps[i] = (float)((x == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.42 - 0.5 * cos(2 * M_PI * i / (NT - 1)) + 0.08 * cos(4 * M_PI * i / (NT - 1)))); sum += taps[i]; }
    for (auto& t : taps) t = (float)(t / sum);
    void *sine = nullptr, *vfo = nullptr;
    if (qdsp_hip_sine_cf32_create(&sine, 0, cosf(0.1f), sinf(0.1f), n)) return 1;
    if (qdsp_hip_xlate_fir_decim_cf32_create(&vfo, 0, taps.data(), NT, 1, M, cosf(-0.05f), sinf(-0.05f), n)) return 1;
    void *buf[2], *out[2];
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&buf[i], (size_t)n * 8)); CK(hipMalloc(&out[i], (size_t)(n / M + 64) * 8)); }
    hipStream_t s[2];
    hipEvent_t ready[2], freed[2], ordv, ords;
    for (int i = 0; i < 2; i++) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ready[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&freed[i], hipEventDisableTiming)); }
    CK(hipEventCreateWithFlags(&ordv, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ords, hipEventDisableTiming));
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < nblk; k++) {
                const int p = k & 1;
                if (mode == 0) {
                    qdsp_hip_sine_cf32_generate_dev(sine, n, buf[p], s[0]);
                    qdsp_hip_xlate_fir_decim_cf32_process_dev(vfo, buf[p], n, out[p], s[0]);
                } else if (mode == 1) {
                    qdsp_hip_sine_cf32_generate_dev(sine, n, buf[p], s[p]);
                    if (k) CK(hipStreamWaitEvent(s[p], ordv, 0));
                    qdsp_hip_xlate_fir_decim_cf32_process_dev(vfo, buf[p], n, out[p], s[p]);
                    CK(hipEventRecord(ordv, s[p]));
                } else {
                    if (k >= 2) CK(hipStreamWaitEvent(s[0], freed[p], 0));      // the VFO has read this buffer
                    qdsp_hip_sine_cf32_generate_dev(sine, n, buf[p], s[0]);
                    CK(hipEventRecord(ready[p], s[0]));
                    CK(hipStreamWaitEvent(s[1], ready[p], 0));
                    qdsp_hip_xlate_fir_decim_cf32_process_dev(vfo, buf[p], n, out[p], s[1]);
                    CK(hipEventRecord(freed[p], s[1]));
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%-6s %d blocks of %d: %.1f us per block = %.1f Gs/s\n", mode == 0 ? "one" : mode == 1 ? "lanes" : "stages", nblk, n, us / nblk, n / (us / nblk) / 1e3);
        }
    }
    return 0;
}
