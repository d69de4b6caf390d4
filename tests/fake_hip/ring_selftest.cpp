// TEST-ONLY driver of qdsp_amd/csrc/ring.cpp against the fake HIP runtime + fake one-rank RCCL (fake_hip_runtime.cpp), built with
// -fsanitize=address,undefined: one rank as its own ring neighbour, 200 steps with one exchange posted ahead (what
// qdsp_amd/sharding.py RingStream and graph_check `shard` do), every halo and prev-halo pointer checked against what was sent.
#include "../../include/qdsp_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                            \
    do {                                                                    \
        if (!(x)) { printf("FAIL line %d: %s\n", __LINE__, #x); return 1; } \
    } while (0)

int main() {
    const int H = 255 * 8;
    CHECK(qdsp_hip_ring_available() == 1);
    char id[QDSP_HIP_RING_ID_BYTES];
    CHECK(qdsp_hip_ring_unique_id(id) == 0);
    void* ring = nullptr;
    CHECK(qdsp_hip_ring_create(&ring, 5, 0, 1, id, H) == QDSP_HIP_ENODEV);     // one device only
    CHECK(qdsp_hip_ring_create(&ring, 0, 1, 1, id, H) == QDSP_HIP_EINVAL);     // rank outside the world
    CHECK(qdsp_hip_ring_create(&ring, 0, 0, 1, id, H) == 0 && ring);
    int cr = -1, rk = -1, dv = -1, ver = -1;
    CHECK(qdsp_hip_ring_info(ring, &cr, &rk, &dv, &ver) == 0 && cr == 1 && rk == 0 && dv == 0 && ver > 0);
    CHECK(qdsp_hip_ring_set_timing(ring, 1) == 0);
    const void *halo = nullptr, *prev = nullptr;
    CHECK(qdsp_hip_ring_complete(ring, nullptr, &halo, &prev) == QDSP_HIP_EINVAL);   // nothing posted
    const int steps = 200;
    char* tails = static_cast<char*>(malloc((size_t)(steps + 2) * H));
    for (int s = 0; s < steps + 2; s++) memset(tails + (size_t)s * H, 1 + s % 250, H);
    auto tail = [&](int s) { return tails + (size_t)s * H; };
    // (a) one exchange ahead: step s reads post s and posts s + 1 before "its kernel"
    CHECK(qdsp_hip_ring_post(ring, tail(0), nullptr) == 0);
    for (int s = 0; s < steps; s++) {
        CHECK(qdsp_hip_ring_complete(ring, nullptr, &halo, &prev) == 0);
        CHECK(memcmp(halo, tail(s), H) == 0);                                              // what arrived with post s
        if (s == 0) { for (int i = 0; i < H; i++) CHECK(static_cast<const char*>(prev)[i] == 0); }   // zeros before the first
        else CHECK(memcmp(prev, tail(s - 1), H) == 0);                                     // what arrived a step earlier
        CHECK(qdsp_hip_ring_post(ring, tail(s + 1), nullptr) == 0);
    }
    CHECK(qdsp_hip_ring_drain(ring) == 0);
    // (b) two ahead: the third post is refused, and neither allowed post lands in a buffer the last complete handed out
    CHECK(qdsp_hip_ring_post(ring, tail(0), nullptr) == 0);
    CHECK(qdsp_hip_ring_post(ring, tail(1), nullptr) == 0);
    CHECK(qdsp_hip_ring_post(ring, tail(2), nullptr) == QDSP_HIP_EINVAL);
    for (int s = 0; s < 50; s++) {
        CHECK(qdsp_hip_ring_complete(ring, nullptr, &halo, &prev) == 0);
        CHECK(memcmp(halo, tail(s), H) == 0);
        CHECK(qdsp_hip_ring_post(ring, tail(s + 2), nullptr) == 0);       // posts s + 1 and s + 2 are now outstanding
        CHECK(qdsp_hip_ring_post(ring, tail(0), nullptr) == QDSP_HIP_EINVAL);
        CHECK(memcmp(halo, tail(s), H) == 0);                              // still intact behind two posts
        if (s) CHECK(memcmp(prev, tail(s - 1), H) == 0);
    }
    CHECK(qdsp_hip_ring_drain(ring) == 0);
    double mean = 0, mx = 0;
    long long n = 0;
    CHECK(qdsp_hip_ring_exchange_us(ring, &mean, &mx, &n) == 0 && n > 0 && mean > 0 && mx >= mean);
    qdsp_hip_ring_destroy(ring);
    qdsp_hip_ring_destroy(nullptr);
    free(tails);
    printf("ring ok (%lld exchanges timed)\n", n);
    return 0;
}
