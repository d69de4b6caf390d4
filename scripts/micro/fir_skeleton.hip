// fir_skeleton.hip (round 4) -- would a 2048-point overlap-save FIR beat the 4096-point one?  Memory skeletons of both, WITH the overlap:
// a 256-thread workgroup loads a segment of F complex samples (F = 4096: 32 KB, F = 2048: 16 KB) that starts every L = F - 256 samples
// (256 taps: the first 256 elements of a segment are history), does `work` x 4 PER dependent-free v_fma_f32 per lane, and stores the L valid
// outputs.  2^27 input samples in all, as bench.py's fir256.  work: 30 at F = 4096 is the 4096-point kernel's arithmetic (burst_copy.hip);
// a 2048-point transform pair does 11/12 of it per sample: 55 at F = 2048 (half the values per lane).
//   one-shot: one workgroup per segment;  persistent: 256 x cap workgroups, the next segment's loads issued before the work (register prefetch)
//   hipcc --offload-arch=gfx950 -O3 -o fir_skeleton fir_skeleton.hip && ./fir_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PER> __device__ __forceinline__ void work_on(float4 (&v)[PER], int work, float ka, float kb) {
    for (int w = 0; w < work; w++) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            v[u].x = fmaf(v[u].x, ka, kb);
            v[u].y = fmaf(v[u].y, ka, kb);
            v[u].z = fmaf(v[u].z, ka, kb);
            v[u].w = fmaf(v[u].w, ka, kb);
        }
    }
}

// PER float4 per lane = PER * 512 samples per segment; stride L4 = (F - 256) / 2 float4
template <int PER, bool PREFETCH>
__global__ __launch_bounds__(256) void seg(const float4* __restrict__ in, float4* __restrict__ out, int nseg, int work, float ka, float kb) {
    extern __shared__ char occupancy_cap[];
    constexpr int L4 = PER * 256 - 128;
    const int t = threadIdx.x;
    float4 v[PER], n[PER];
    int s = blockIdx.x;
    if (s >= nseg) return;
    if (PREFETCH) {
        const float4* p = in + (long long)s * L4 + t;
#pragma unroll
        for (int u = 0; u < PER; u++) v[u] = p[u * 256];
    }
    for (; s < nseg; s += gridDim.x) {
        if (PREFETCH) {
            const int sn = s + gridDim.x;
            const float4* p = in + (long long)(sn < nseg ? sn : s) * L4 + t;
#pragma unroll
            for (int u = 0; u < PER; u++) n[u] = p[u * 256];
            asm volatile("" ::: "memory");
        } else {
            const float4* p = in + (long long)s * L4 + t;
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = p[u * 256];
        }
        work_on<PER>(v, work, ka, kb);
        float4* q = out + (long long)s * L4 + t - 128;      // element e of the segment -> output e - 128 (the first 128 float4 are history)
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (u > 0 || t >= 128) q[u * 256] = v[u];
        if (PREFETCH) {
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = n[u];
        }
    }
}

__global__ void fill(float4* a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = make_float4(1.0f, 0.5f, 0.25f, 0.125f);
}

static float4 *A, *B;
static const long long N = 1ll << 26;   // float4 elements = 2^27 samples
static hipEvent_t e0, e1;

template <int PER, bool PF> float time_one(int grid, int cap, int work) {
    constexpr int L4 = PER * 256 - 128;
    const int nseg = (int)((N - 128) / L4);
    const int g = grid ? grid : nseg;
    const size_t lds = cap >= 8 ? 0 : (160 * 1024 / cap) - 1024;
    (void)hipFuncSetAttribute((const void*)seg<PER, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((seg<PER, PF>), dim3(g), dim3(256), lds, 0, A, B, nseg, work, 0.999f, 0.001f);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((seg<PER, PF>), dim3(g), dim3(256), lds, 0, A, B, nseg, work, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int PER> void sweep(int w_fir) {
    for (int work : {0, w_fir}) {
        for (int cap : {8, 4}) {
            printf("F %4d (%2d KB in, overlap x%.3f) work %2d cap %d WG/CU: one-shot %.4f ms", PER * 512, PER * 4, PER * 256.0 / (PER * 256 - 128), work, cap,
                   time_one<PER, false>(0, cap, work));
            for (int q : {1, 4}) printf("   persistent x%d: plain %.4f prefetch %.4f", q, time_one<PER, false>(256 * cap * q, cap, work), time_one<PER, true>(256 * cap * q, cap, work));
            printf("\n");
            fflush(stdout);
        }
    }
}

int main() {
    (void)hipMalloc(&A, (N + 4096) * 16);
    (void)hipMalloc(&B, (N + 4096) * 16);
    hipLaunchKernelGGL(fill, dim3((unsigned)(N / 256)), dim3(256), 0, 0, A, N);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) (void)time_one<8, false>(1024, 4, 0);   // settle the clocks
    for (int rep = 0; rep < 2; rep++) {
        sweep<8>(30);
        sweep<4>(55);
        sweep<2>(100);   // 1024-point segments as a workgroup would move them (x1.33 overlap)
    }
    return 0;
}
