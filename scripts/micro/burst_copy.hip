// burst_copy.hip -- memory skeletons of a "load a segment, work on it, store it" kernel (1 GiB in, 1 GiB out), to find out
// why every 16 B/sample filter kernel of this library sits at 0.40-0.43 ms per 2^27 samples while a one-float4-per-lane
// copy takes 0.343 ms (profiles/r01_micro_copy_bw.txt).  Each 256-thread workgroup handles segments of PER x 4 KB.
//   MODE 0  burst:     load the whole segment, work, store the whole segment           (fir_fft_kernel's shape at PER = 8)
//   MODE 1  prefetch:  the next segment's loads are issued before the work on this one (register double buffer)
//   MODE 2  trickle:   the next segment's loads AND the previous segment's stores are issued in four pieces between the
//                      four quarters of the work (smooth request stream)
// work = iterations of 4*PER independent v_fma_f32 per lane between load and store (0: pure copy; 30 at PER = 8 is about
// the 4300 VALU cycles per wave fir_fft_kernel<1> spends on a 4096-point segment).  `cap` limits resident workgroups per
// CU through a dynamic LDS allocation; grid = 0 launches one workgroup per segment, otherwise a persistent grid.
//   hipcc --offload-arch=gfx950 -O3 -o burst_copy burst_copy.hip && ./burst_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PER>
__device__ __forceinline__ void work_on(float4 (&v)[PER], int from, int to, float ka, float kb) {
    for (int w = from; w < to; w++) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            v[u].x = fmaf(v[u].x, ka, kb);
            v[u].y = fmaf(v[u].y, ka, kb);
            v[u].z = fmaf(v[u].z, ka, kb);
            v[u].w = fmaf(v[u].w, ka, kb);
        }
    }
}

template <int PER, int MODE>
__global__ __launch_bounds__(256) void seg_copy(const float4* __restrict__ in, float4* __restrict__ out, int nseg, int work, float ka, float kb) {
    extern __shared__ char occupancy_cap[];
    const int t = threadIdx.x;
    if (MODE == 0) {
        for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
            const float4* p = in + (long long)s * PER * 256 + t;
            float4* q = out + (long long)s * PER * 256 + t;
            float4 v[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = p[u * 256];
            work_on<PER>(v, 0, work, ka, kb);
#pragma unroll
            for (int u = 0; u < PER; u++) q[u * 256] = v[u];
        }
    } else if (MODE == 1) {
        int s = blockIdx.x;
        if (s >= nseg) return;
        float4 v[PER], n[PER];
        {
            const float4* p = in + (long long)s * PER * 256 + t;
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = p[u * 256];
        }
        for (; s < nseg; s += gridDim.x) {
            const int sn = s + gridDim.x;
            if (sn < nseg) {
                const float4* p = in + (long long)sn * PER * 256 + t;
#pragma unroll
                for (int u = 0; u < PER; u++) n[u] = p[u * 256];
            }
            asm volatile("" ::: "memory");
            work_on<PER>(v, 0, work, ka, kb);
            float4* q = out + (long long)s * PER * 256 + t;
#pragma unroll
            for (int u = 0; u < PER; u++) q[u * 256] = v[u];
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = n[u];
        }
    } else {
        constexpr int Q = PER / 4 > 0 ? PER / 4 : 1;   // loads / stores per quarter
        constexpr int NQ = PER / Q;
        int s = blockIdx.x;
        if (s >= nseg) return;
        float4 v[PER], n[PER], o[PER];
        {
            const float4* p = in + (long long)s * PER * 256 + t;
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = p[u * 256];
        }
        bool have_o = false;
        long long so = 0;
        for (; s < nseg; s += gridDim.x) {
            const int sn = s + gridDim.x;
            const float4* p = in + (long long)(sn < nseg ? sn : s) * PER * 256 + t;
            float4* q = out + so * PER * 256 + t;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
#pragma unroll
                for (int u = qq * Q; u < (qq + 1) * Q; u++) n[u] = p[u * 256];
                if (have_o) {
#pragma unroll
                    for (int u = qq * Q; u < (qq + 1) * Q; u++) q[u * 256] = o[u];
                }
                asm volatile("" ::: "memory");
                work_on<PER>(v, work * qq / NQ, work * (qq + 1) / NQ, ka, kb);
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int u = 0; u < PER; u++) { o[u] = v[u]; v[u] = n[u]; }
            have_o = true;
            so = s;
        }
        float4* q = out + so * PER * 256 + t;
#pragma unroll
        for (int u = 0; u < PER; u++) q[u * 256] = o[u];
    }
}

__global__ void fill(float4* a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = make_float4(1.0f, 0.5f, 0.25f, 0.125f);
}

static float4 *A, *B;
static const long long N = 1ll << 26;   // float4 elements: 1 GiB each way
static hipEvent_t e0, e1;

template <int PER, int MODE> float time_one(int grid, int cap, int work) {
    const int nseg = (int)(N / (PER * 256));
    const int g = grid ? grid : nseg;
    const size_t lds = cap >= 8 ? 0 : (160 * 1024 / cap) - 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)seg_copy<PER, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((seg_copy<PER, MODE>), dim3(g), dim3(256), lds, 0, A, B, nseg, work, 0.999f, 0.001f);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((seg_copy<PER, MODE>), dim3(g), dim3(256), lds, 0, A, B, nseg, work, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int PER> void sweep() {
    const int works[3] = {0, 15 * 8 / PER, 30 * 8 / PER};   // same VALU work per byte at every PER
    for (int cap : {8, 4, 2}) {
        for (int grid : {0, 256 * cap}) {
            for (int wi = 0; wi < 3; wi++) {
                const int work = works[wi];
                const float a = time_one<PER, 0>(grid, cap, work);
                float b = -1, c = -1;
                if (grid) {
                    b = time_one<PER, 1>(grid, cap, work);
                    c = time_one<PER, 2>(grid, cap, work);
                }
                printf("PER %d (%2d KB/segment) cap %d WG/CU %-10s work %3d : burst %.3f ms", PER, PER * 4, cap, grid ? "persistent" : "one-shot",
                       work, a);
                if (grid) printf("  prefetch %.3f  trickle %.3f", b, c);
                printf("\n");
                fflush(stdout);
            }
        }
    }
}

int main() {
    (void)hipMalloc(&A, N * 16);
    (void)hipMalloc(&B, N * 16);
    hipLaunchKernelGGL(fill, dim3((unsigned)(N / 256)), dim3(256), 0, 0, A, N);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // settle the clocks
    for (int i = 0; i < 50; i++) (void)time_one<8, 0>(1024, 4, 0);
    sweep<1>();
    sweep<2>();
    sweep<4>();
    sweep<8>();
    return 0;
}
