// Microbenchmark (round 4): which STORE SHAPE should the oversampled channelizer (chan_uniform_kernel<.,8>) use?
// Every wave writes tiles of T output times x 64 channel rows (8 bytes per sample), with `work` dependent packed FMAs per lane
// between two store instructions (the kernel's arithmetic: ~35 VALU instructions per store), 12 waves per CU, non-temporal stores:
//   S0  T = 16: 16 x 8-byte stores,  each 4 rows x 128-byte runs   (the kernel of rounds 2-4)
//   S1  T = 32: 32 x 8-byte stores,  each 2 rows x 256-byte runs   (two tiles paired with v_permlane16_swap)
//   S2  T = 16:  8 x 16-byte stores, each 8 rows x 128-byte runs   (adjacent lanes traded)
//   S3  T = 32: 16 x 16-byte stores, each 4 rows x 256-byte runs   (even / odd output times in one lane)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/css scripts/micro/chan_store_shapes.hip && /tmp/css
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void wr(float2* out, long long stride, int ntiles, int nwaves, int work, int nostore) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    constexpr int T = (SHAPE == 1 || SHAPE == 3) ? 32 : 16;
    v2f a = {1.0f + l * 1e-3f, 0.5f}, b = {0.999f, 1e-3f}, c = {1e-4f, -1e-4f};
    for (int t = gw; t < ntiles; t += nwaves) {
        const long long n0 = (long long)t * T;
        constexpr int NST = (SHAPE == 0) ? 16 : (SHAPE == 1) ? 32 : (SHAPE == 2) ? 8 : 16;
#pragma unroll 4
        for (int k = 0; k < NST; k++) {
            for (int w = 0; w < work; w++) a = __builtin_elementwise_fma(a, b, c);
            if (SHAPE == 0) {          // rows k, k+16, k+32, k+48; 16 lanes per row
                v2f* p = reinterpret_cast<v2f*>(out) + (size_t)(k + 16 * (l >> 4)) * stride + n0 + (l & 15);
                if (!nostore || a.x == 1.2345e30f) __builtin_nontemporal_store(a, p);
            } else if (SHAPE == 1) {   // rows k, k+32; 32 lanes per row
                v2f* p = reinterpret_cast<v2f*>(out) + (size_t)(k + 32 * (l >> 5)) * stride + n0 + (l & 31);
                if (!nostore || a.x == 1.2345e30f) __builtin_nontemporal_store(a, p);
            } else if (SHAPE == 2) {   // rows k + 8 j (j = l >> 3); 8 lanes x 16 bytes per row
                v4f* p = reinterpret_cast<v4f*>(reinterpret_cast<v2f*>(out) + (size_t)(k + 8 * (l >> 3)) * stride + n0 + 2 * (l & 7));
                if (!nostore || a.x == 1.2345e30f) __builtin_nontemporal_store((v4f){a.x, a.y, a.y, a.x}, p);
            } else {                   // rows k + 16 j (j = l >> 4); 16 lanes x 16 bytes per row
                v4f* p = reinterpret_cast<v4f*>(reinterpret_cast<v2f*>(out) + (size_t)(k + 16 * (l >> 4)) * stride + n0 + 2 * (l & 15));
                if (!nostore || a.x == 1.2345e30f) __builtin_nontemporal_store((v4f){a.x, a.y, a.y, a.x}, p);
            }
        }
    }
}

template <int SHAPE> void run(float2* d, long long stride, long long nout, int work, int nostore, int wgpc) {
    constexpr int T = (SHAPE == 1 || SHAPE == 3) ? 32 : 16;
    const int ntiles = (int)(nout / T);
    const int nwg = 256 * wgpc, nwaves = nwg * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(wr<SHAPE>, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves, work, nostore);
    hipEventRecord(e0);
    const int it = 6;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL(wr<SHAPE>, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves, work, nostore);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("S%d work %3d %s wg/CU %2d: %.3f ms  %.0f GB/s\n", SHAPE, work, nostore ? "NO stores" : "stores   ", wgpc, ms, nostore ? 0.0 : 64.0 * nout * 8 / ms / 1e6);
}

int main() {
    const long long nout = 1LL << 24, stride = nout + 32;
    float2* d;
    if (hipMalloc(&d, 64 * stride * sizeof(float2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int wgpc : {48}) {
        for (int work : {0, 8, 16, 24, 32}) {
            // per-store work differs per shape so that the work per OUTPUT SAMPLE is the same: shapes with 16-byte stores do twice the work per store
            run<0>(d, stride, nout, work, 0, wgpc);
            run<1>(d, stride, nout, work, 0, wgpc);
            run<2>(d, stride, nout, 2 * work, 0, wgpc);
            run<3>(d, stride, nout, 2 * work, 0, wgpc);
            if (work) run<0>(d, stride, nout, work, 1, wgpc);
        }
    }
    return 0;
}
