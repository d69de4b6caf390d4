// Microbenchmark: HBM write bandwidth when every wave writes RUN contiguous bytes into each of 64
// output rows (the channelizer's store pattern), as a function of RUN.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ws scripts/micro/write_streams.hip && /tmp/ws
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int RUN_ELEMS>   // float2 elements per row per wave tile (16 -> 128 B)
__global__ __launch_bounds__(256) void wr(float2* out, long long stride, int ntiles, int nwaves) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    constexpr int ROWS_PER_STORE = 64 / RUN_ELEMS > 0 ? 64 / RUN_ELEMS : 1;      // rows covered by one store instruction
    constexpr int STORES_PER_ROWSET = RUN_ELEMS > 64 ? RUN_ELEMS / 64 : 1;
    for (int t = gw; t < ntiles; t += nwaves) {
        const long long n0 = (long long)t * RUN_ELEMS;
        if (RUN_ELEMS <= 64) {
            const int row_in = l / RUN_ELEMS, col = l % RUN_ELEMS;
#pragma unroll
            for (int k = 0; k < 64 / ROWS_PER_STORE; k++) {
                const int c = k * ROWS_PER_STORE + row_in;
                out[(size_t)c * stride + n0 + col] = make_float2((float)t, (float)c);
            }
        } else {
            for (int c = 0; c < 64; c++)
#pragma unroll
                for (int s = 0; s < STORES_PER_ROWSET; s++) out[(size_t)c * stride + n0 + s * 64 + l] = make_float2((float)t, (float)c);
        }
    }
}

// the same 128-byte runs, but adjacent lanes go to different rows (lane -> row l & 3, column l >> 2):
// what a quad-wise DPP butterfly leaves behind
__global__ __launch_bounds__(256) void wr_quad(float2* out, long long stride, int ntiles, int nwaves) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    for (int t = gw; t < ntiles; t += nwaves) {
        const long long n0 = (long long)t * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int c = k + 16 * (l & 3);
            out[(size_t)c * stride + n0 + (l >> 2)] = make_float2((float)t, (float)c);
        }
    }
}

// memory skeleton of the channelizer: per wave tile read 19 rows of 64 consecutive float2 (tiles overlap
// by 3 rows) and write 16 x (4 rows x 128 B); ROWS_IN = 16 reads without the overlap
template <int ROWS_IN>
__global__ __launch_bounds__(256) void skel(const float2* in, float2* out, long long stride, int ntiles, int nwaves) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    for (int t = gw; t < ntiles; t += nwaves) {
        const float2* src = in + (long long)t * 1024 + l;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < ROWS_IN; r++) { const float2 v = src[64 * r]; acc.x += v.x; acc.y += v.y; }
        const long long n0 = (long long)t * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int c = k + 16 * (l & 3);
            out[(size_t)c * stride + n0 + (l >> 2)] = make_float2(acc.x + k, acc.y);
        }
    }
}

template <int RUN_ELEMS> void run(float2* d, long long stride, const char* name) {
    const int ntiles = (int)(stride / RUN_ELEMS);
    const int nwg = 256 * 12, nwaves = nwg * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(wr<RUN_ELEMS>, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL(wr<RUN_ELEMS>, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-28s run %5d B x 64 rows: %.3f ms  %.0f GB/s\n", name, RUN_ELEMS * 8, ms, 64.0 * stride * 8 / ms / 1e6);
}

__global__ void lin(float4* out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = make_float4(1, 2, 3, 4);
}

int main(int argc, char** argv) {
    // 2^21 outputs per row = 16 MiB rows, 1 GiB total (the channelizer at decimation 64); pass 24 for 128 MiB rows, 8 GiB (decimation 8)
    const long long stride = 1ll << (argc > 1 ? atoi(argv[1]) : 21);
    float2* d; hipMalloc(&d, 64 * stride * 8);
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL(lin, dim3(4096), dim3(256), 0, 0, (float4*)d, 64 * stride / 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(lin, dim3(4096), dim3(256), 0, 0, (float4*)d, 64 * stride / 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("linear 16 B/lane fill 1 GiB: %.3f ms %.0f GB/s\n", ms / 10, 64.0 * stride * 8 / (ms / 10) / 1e6);
    {
        const int ntiles = (int)(stride / 16), nwg = 256 * 12, nwaves = nwg * 4;
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(wr_quad, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves);
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(wr_quad, dim3(nwg), dim3(256), 0, 0, d, stride, ntiles, nwaves);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("quad-interleaved lanes, 128 B runs x 64 rows: %.3f ms %.0f GB/s\n", ms / 10, 64.0 * stride * 8 / (ms / 10) / 1e6);
    }
    {
        float2* din; hipMalloc(&din, 64 * stride * 8 + 8192);
        hipMemset(din, 0, 64 * stride * 8 + 8192);
        const int ntiles = (int)(stride / 16);
        for (int wpc = 8; wpc <= 512; wpc *= 4) {
            const int nwg = 256 * wpc / 4 * 4 / 4 * 1, nwaves = nwg * 4;
            for (int i = 0; i < 3; i++) hipLaunchKernelGGL(skel<19>, dim3(nwg), dim3(256), 0, 0, din, d, stride, ntiles, nwaves);
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(skel<19>, dim3(nwg), dim3(256), 0, 0, din, d, stride, ntiles, nwaves);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("skeleton 19 rows in, nwg %d: %.3f ms  (2 GiB algorithmic: %.0f GB/s)\n", nwg, ms / 10, 2.0 * 64.0 * stride * 8 / (ms / 10) / 1e6);
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(skel<16>, dim3(nwg), dim3(256), 0, 0, din, d, stride, ntiles, nwaves);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("skeleton 16 rows in, nwg %d: %.3f ms  (2 GiB algorithmic: %.0f GB/s)\n", nwg, ms / 10, 2.0 * 64.0 * stride * 8 / (ms / 10) / 1e6);
        }
    }
    run<16>(d, stride, "64 rows");
    run<32>(d, stride, "64 rows");
    run<64>(d, stride, "64 rows");
    run<128>(d, stride, "64 rows");
    run<256>(d, stride, "64 rows");
    run<512>(d, stride, "64 rows");
    return 0;
}
