// stream_read.hip -- read bandwidth of "every wave streams its own segments" access patterns on gfx950, against the
// dense grid-stride front.  A wave reads segments b = wave + k * nwaves of SEG bytes each (rows of 512 B: one
// global_load_dwordx2 per lane), DEPTH rows in flight at a time; everything is summed so nothing is optimised away.
// hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int DEPTH>
__global__ __launch_bounds__(256) void k_stream(const float2* __restrict__ in, float* out, long long nseg, int rows, int stride_rows, int nwaves) {
    const int l = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float2 acc = make_float2(0.f, 0.f);
    for (long long b = wave; b < nseg; b += nwaves) {
        const float2* p = in + b * (long long)stride_rows * 64 + l;
        for (int r0 = 0; r0 < rows; r0 += DEPTH) {
            float2 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; i++) v[i] = p[(long long)(r0 + i) * 64];
#pragma unroll
            for (int i = 0; i < DEPTH; i++) { acc.x += v[i].x; acc.y += v[i].y; }
        }
    }
    if (acc.x == 12345.f) out[wave] = acc.x + acc.y;
}

// the polyphase decimator's shape: segments of 4096 samples every `stride` samples (3848: a 64-byte-odd start for
// every other segment), 64 rows in flight, 481 of 512 results stored per segment
__global__ __launch_bounds__(256, 2) void k_pfb_like(const float2* __restrict__ in, float2* __restrict__ outp, long long nseg, int stride, int nwaves, int store) {
    const int l = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long b = wave; b < nseg; b += nwaves) {
        const float2* p = in + b * (long long)stride + l;
        float2 v[64];
#pragma unroll
        for (int i = 0; i < 64; i++) v[i] = p[(long long)i * 64];
        float2 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[i] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; j++) { acc[i].x += v[8 * i + j].x; acc[i].y += v[8 * i + j].y; }
        }
        if (store == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) { const int ap = l + 64 * i; if (ap >= 31) outp[b * 481 + ap - 31] = acc[i]; }
        } else if (store == 2) {       // 480 results per segment: every segment's output starts on a 128-byte line
#pragma unroll
            for (int i = 0; i < 8; i++) { const int ap = l + 64 * i; if (ap >= 32) outp[b * 480 + ap - 32] = acc[i]; }
        } else if (store == 3) {       // the same, non-temporal
#pragma unroll
            for (int i = 0; i < 8; i++) { const int ap = l + 64 * i; if (ap >= 32) { typedef float v2 __attribute__((ext_vector_type(2))); __builtin_nontemporal_store((v2){acc[i].x, acc[i].y}, reinterpret_cast<v2*>(&outp[b * 480 + ap - 32])); } }
        } else if (store == 4) {       // 16-byte stores: lanes pair up (two results per lane), 4 instructions
#pragma unroll
            for (int i = 0; i < 4; i++) { const int ap = 2 * l + 128 * i; if (ap >= 32) *reinterpret_cast<float4*>(&outp[b * 480 + ap - 32]) = make_float4(acc[2*i].x, acc[2*i].y, acc[2*i+1].x, acc[2*i+1].y); }
        } else if (acc[0].x == 12345.f) outp[wave] = acc[0];
    }
}
__global__ void k_fill(float* p, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
    }
}

__global__ __launch_bounds__(256) void k_dense(const float4* __restrict__ in, float* out, long long n4) {
    float4 acc = make_float4(0, 0, 0, 0);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) { const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.f) out[blockIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH> float run_stream(const float2* in, float* out, long long bytes, int seg_bytes, int waves_per_cu) {
    const int rows = seg_bytes / 512, nwaves = 256 * waves_per_cu;
    const long long nseg = bytes / seg_bytes;
    if (rows % DEPTH != 0 || nseg * (long long)seg_bytes > bytes) return -1.0f;     // (a batch must not run past its segment)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream<DEPTH>, dim3(nwaves / 4), dim3(256), 0, 0, in, out, nseg, rows, rows, nwaves);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const long long bytes = 1LL << 30;
    float2* in;
    float* out;
    (void)hipMalloc(&in, bytes);
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(in, 0, bytes);
    float2* outp;
    (void)hipMalloc(&outp, bytes / 8 + (1 << 20));
    for (int pass = 1; pass < 2; pass++) {
        if (pass == 1) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float*>(in), bytes / 4); (void)hipDeviceSynchronize(); }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int stride : {3848, 3840})
            for (int store : {1, 2, 3, 4})
                for (int wpc : {8}) {
                    const long long nseg = (bytes / 8 - 4096) / stride;
                    float best = 1e9f;
                    for (int rep = 0; rep < 6; rep++) {
                        (void)hipEventRecord(e0);
                        hipLaunchKernelGGL(k_pfb_like, dim3(256 * wpc / 4), dim3(256), 0, 0, in, outp, nseg, stride, 256 * wpc, store);
                        (void)hipEventRecord(e1);
                        (void)hipEventSynchronize(e1);
                        float ms;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        if (rep >= 2 && ms < best) best = ms;
                    }
                    printf("%s data, pfb-like: stride %d samples, store %d, %d waves/CU: %.3f ms  %.0f GB/s read (of the %lld segments)\n", pass ? "random" : "zero", stride,
                           store, wpc, best, nseg * 32768.0 / best / 1e6, nseg);
                }
    }
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int grid : {2048, 8192}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_dense, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(in), out, bytes / 16);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("dense grid-stride float4, grid %5d: %.3f ms  %.0f GB/s\n", grid, best, bytes / best / 1e6);
        }
    }
    for (int seg : {8192, 16384, 32768, 65536, 262144})
        for (int wpc : {4, 8, 16, 32}) {
            const float a = run_stream<8>(in, out, bytes, seg, wpc), b = run_stream<16>(in, out, bytes, seg, wpc), c = run_stream<64>(in, out, bytes, seg, wpc < 16 ? wpc : 8);
            printf("segment %6d B, %2d waves/CU: depth 8 %.3f ms %5.0f GB/s | depth 16 %.3f ms %5.0f GB/s | depth 64 (<=8 w/CU) %.3f ms %5.0f GB/s\n", seg, wpc, a,
                   a > 0 ? bytes / a / 1e6 : 0.0, b, b > 0 ? bytes / b / 1e6 : 0.0, c, c > 0 ? bytes / c / 1e6 : 0.0);
        }
    return 0;
}
