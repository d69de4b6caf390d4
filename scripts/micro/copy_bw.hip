// Microbenchmark: what a plain device-to-device copy of 1 GiB reaches on this GPU (read 1 GiB + write
// 1 GiB), the practical ceiling for the 16 B/sample FIR and channelizer kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cb scripts/micro/copy_bw.hip && /tmp/cb
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(256) void copy16(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    for (long long base = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n; base += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = in[base + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) out[base + u * 256] = v[u];
    }
}
// one float4 per lane, no loop: as many blocks as it takes
__global__ __launch_bounds__(256) void copy16_oneshot(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
template <int PER> __global__ __launch_bounds__(256) void copy16_few(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long base = (long long)blockIdx.x * 256 * PER + threadIdx.x;
    float4 v[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) if (base + u * 256 < n) v[u] = in[base + u * 256];
#pragma unroll
    for (int u = 0; u < PER; u++) if (base + u * 256 < n) out[base + u * 256] = v[u];
}

__global__ __launch_bounds__(256) void read16(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long long base = (long long)blockIdx.x * 256 * 4 + threadIdx.x; base < n; base += stride) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const float4 v = in[base + u * 256]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (acc.x == 12345.f) out[threadIdx.x] = acc;
}

int main() {
    const long long n = 1ll << 26;   // float4 elements = 1 GiB
    float4 *a, *b;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(copy16<4>, dim3(4096), dim3(256), 0, 0, a, b, n);
    for (int grid = 1024; grid <= 65536; grid *= 2) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy16<4>, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("copy16 unroll 4 grid %6d: %.3f ms  %.0f GB/s (read+write)\n", grid, ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    }
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy16<8>, dim3(8192), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy16 unroll 8 grid   8192: %.3f ms  %.0f GB/s\n", ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy16_oneshot, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy16 one float4 per lane, %lld blocks: %.3f ms  %.0f GB/s\n", n / 256, ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy16_few<2>, dim3((unsigned)(n / 512)), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy16 two float4 per lane, %lld blocks: %.3f ms  %.0f GB/s\n", n / 512, ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy16_few<4>, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy16 four float4 per lane, %lld blocks: %.3f ms  %.0f GB/s\n", n / 1024, ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D 1 GiB:         %.3f ms  %.0f GB/s\n", ms / 10, 2.0 * n * 16 / (ms / 10) / 1e6);
    for (int grid = 2048; grid <= 16384; grid *= 2) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("read16 grid %6d:          %.3f ms  %.0f GB/s (read only)\n", grid, ms / 10, 1.0 * n * 16 / (ms / 10) / 1e6);
    }
    return 0;
}
