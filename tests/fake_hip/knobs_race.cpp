// TEST-ONLY driver for qdsp_amd/csrc/knobs.cpp under -fsanitize=thread: readers call knob() (what every process* call does, 20-25
// times) while other threads rebuild the table through knobs_reload() (tests and tuning scripts do, via qdsp_hip_reload_env) and
// change the environment between reloads.  A reader must always see ONE consistent snapshot; old snapshots are never freed.
#include "../../qdsp_amd/csrc/knobs.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

int main() {
    std::atomic<bool> stop{false};
    std::atomic<long> bad{0};
    setenv("QDSP_HIP_FFT_WG_PER_CU", "4", 1);
    setenv("QDSP_HIP_PFB_WG_PER_CU", "4", 1);
    qk::knobs_reload();
    std::vector<std::thread> readers;
    for (int t = 0; t < 6; t++) {
        readers.emplace_back([&] {
            while (!stop.load(std::memory_order_relaxed)) {
                // both variables are always set to the SAME value before a reload: a torn table would show two different ones
                const qk::Knobs* s = qk::knobs_snapshot();
                const int a = s->set[qk::K_FFT_WG_PER_CU] ? s->val[qk::K_FFT_WG_PER_CU] : -1;
                const int b = s->set[qk::K_PFB_WG_PER_CU] ? s->val[qk::K_PFB_WG_PER_CU] : -1;
                if (a != b) bad++;
                (void)qk::knob(qk::K_NT, 0);
            }
        });
    }
    std::thread writer([&] {
        for (int i = 0; i < 2000; i++) {
            char v[16];
            snprintf(v, sizeof v, "%d", 1 + i % 7);
            // (setenv while other threads getenv is itself a race in libc: only THIS thread touches the environment, and
            // only knobs_reload() -- serialised by its mutex -- reads it)
            setenv("QDSP_HIP_FFT_WG_PER_CU", v, 1);
            setenv("QDSP_HIP_PFB_WG_PER_CU", v, 1);
            qk::knobs_reload();
        }
    });
    writer.join();
    stop = true;
    for (auto& t : readers) t.join();
    if (bad.load()) { printf("FAIL: %ld torn snapshots\n", bad.load()); return 1; }
    printf("knobs ok\n");
    return 0;
}
