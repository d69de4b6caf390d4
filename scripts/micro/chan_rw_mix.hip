// chan_rw_mix.hip (round 4) -- the oversampled channelizer (M = 8) writes 8.6 GB and reads 1 GiB per 2^27 input samples; its ablations say the
// reads cost 0.33 ms -- as if read at 3 TB/s -- because they go to HBM BETWEEN the writes (profiles/r04_chan_tuning.txt block 4).  Does the
// shape of the read stream matter?  Skeleton: every wave writes tiles of 16 output times x 64 channel rows with 16-byte stores (the shipped
// shape: 8 stores of 8 x 128-byte runs) with `work` dependent packed FMAs per store, and reads per tile, per RD mode:
//   0  nothing                               3  1 KB new per tile, non-temporal
//   1  1 KB of NEW input (what HBM must deliver)        4  8 KB contiguous once per 8 tiles of the wave
//   2  the tile's 3 KB span (as the kernel: overlaps   5  32 KB contiguous once per 32 tiles
//      with the neighbours', L2 absorbs the re-reads)
//   hipcc --offload-arch=gfx950 -O3 -o chan_rw_mix chan_rw_mix.hip && ./chan_rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int RD>
__global__ __launch_bounds__(256, 3) void k(const v4f* __restrict__ in, v2f* out, long long stride, int ntiles, int nwaves, int work) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    v2f a = {1.0f + l * 1e-3f, 0.5f}, b = {0.999f, 1e-3f}, c = {1e-4f, -1e-4f};
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    int it = 0;
    for (int t = gw; t < ntiles; t += nwaves, it++) {
        // ---- reads: tile t's new input is 128 samples = 1 KB at in + 64 t (v4f units)
        if (RD == 1) acc += in[(long long)t * 64 + l];
        if (RD == 3) acc += __builtin_nontemporal_load(in + (long long)t * 64 + l);
        if (RD == 2) {
#pragma unroll
            for (int r = 0; r < 3; r++) acc += in[(long long)t * 64 + 64 * r + l];
        }
        if (RD == 4 && (it & 7) == 0) {
            const long long ch = gw + (long long)(it >> 3) * nwaves;            // 8 KB chunk index: ntiles / 8 of them make up the input
            if (ch < ntiles / 8) {
#pragma unroll
                for (int r = 0; r < 8; r++) acc += in[(ch * 8 + r) * 64 + l];
            }
        }
        if (RD == 5 && (it & 31) == 0) {
            const long long ch = gw + (long long)(it >> 5) * nwaves;
            if (ch < ntiles / 32) {
#pragma unroll 8
                for (int r = 0; r < 32; r++) acc += in[(ch * 32 + r) * 64 + l];
            }
        }
        a.x += acc.x * 1e-30f;
        const long long n0 = (long long)t * 16;
#pragma unroll 4
        for (int kk = 0; kk < 8; kk++) {
            for (int w = 0; w < work; w++) a = __builtin_elementwise_fma(a, b, c);
            v4f* p = reinterpret_cast<v4f*>(out + (size_t)(kk + 8 * (l >> 3)) * stride + n0 + 2 * (l & 7));
            __builtin_nontemporal_store((v4f){a.x, a.y, a.y, a.x}, p);
        }
    }
}

template <int RD> float run(const v4f* in, v2f* d, long long stride, long long nout, int work) {
    const int ntiles = (int)(nout / 16);
    const int nwg = 256 * 48, nwaves = nwg * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<RD>, dim3(nwg), dim3(256), 0, 0, in, d, stride, ntiles, nwaves, work);
    hipEventRecord(e0);
    const int it = 6;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL(k<RD>, dim3(nwg), dim3(256), 0, 0, in, d, stride, ntiles, nwaves, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / it;
}

int main() {
    const long long nout = 1LL << 24, stride = nout + 32;
    v2f* d; v4f* in;
    if (hipMalloc(&d, 64 * stride * sizeof(v2f)) != hipSuccess || hipMalloc(&in, (1LL << 30) + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, (1LL << 30) + (1 << 20));
    for (int rep = 0; rep < 2; rep++)
        for (int work : {0, 48}) {
            printf("work %2d per 16-byte store:  no reads %.3f | 1 KB new per tile %.3f | 3 KB span per tile %.3f | 1 KB nt %.3f | 8 KB per 8 tiles %.3f | 32 KB per 32 tiles %.3f  ms\n", work,
                   run<0>(in, d, stride, nout, work), run<1>(in, d, stride, nout, work), run<2>(in, d, stride, nout, work), run<3>(in, d, stride, nout, work),
                   run<4>(in, d, stride, nout, work), run<5>(in, d, stride, nout, work));
            fflush(stdout);
        }
    return 0;
}
