// burst_wg.hip -- fourth skeleton run: workgroup size x loads per lane, with the cache-policy bits of the loads and stores of a "load segment, store segment"
// kernel (1 GiB in, 1 GiB out; a 256-thread workgroup per PER x 4 KB segment, rows of 4 KB).  Loads and stores go through
// buffer instructions so that the aux field can be set: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1, 18 sc1 nt, 3 sc0 nt.
//   hipcc --offload-arch=gfx950 -O3 -o burst_policy burst_policy.hip && ./burst_policy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int NTH, int PER, int LP, int SP>
__global__ __launch_bounds__(NTH) void seg(const f4* __restrict__ in, f4* __restrict__ out, int nseg) {
    const int t = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 30, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 1 << 30, 0x00020000);
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const unsigned base = (unsigned)s * (PER * NTH * 16u) + t * 16u;
        f4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, base + k * (NTH * 16u), 0, LP));
#pragma unroll
        for (int k = 0; k < PER; k++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v[k]), rs_out, base + k * (NTH * 16u), 0, SP);
    }
}

__global__ void fill(f4* a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = (f4){1.0f, 0.5f, 0.25f, 0.125f};
}

static f4 *A, *B;
static const long long N = 1ll << 26;
static hipEvent_t e0, e1;

template <int NTH, int PER, int LP, int SP> float time_one(int grid) {
    const int nseg = (int)(N / (PER * NTH));
    const int g = grid ? grid : nseg;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((seg<NTH, PER, LP, SP>), dim3(g), dim3(NTH), 0, 0, A, B, nseg);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((seg<NTH, PER, LP, SP>), dim3(g), dim3(NTH), 0, 0, A, B, nseg);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int NTH, int PER> void row() {
    printf("%4d threads x %d loads per lane (%3d KB per workgroup): plain %.3f / %.3f   nt loads + nt stores %.3f / %.3f   nt loads + sc1 stores %.3f / %.3f\n", NTH, PER,
           NTH * PER * 16 / 1024, time_one<NTH, PER, 0, 0>(0), time_one<NTH, PER, 0, 0>(256 * 1024 / NTH), time_one<NTH, PER, 2, 2>(0),
           time_one<NTH, PER, 2, 2>(256 * 1024 / NTH), time_one<NTH, PER, 2, 16>(0), time_one<NTH, PER, 2, 16>(256 * 1024 / NTH));
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&A, N * 16);
    (void)hipMalloc(&B, N * 16);
    hipLaunchKernelGGL(fill, dim3((unsigned)(N / 256)), dim3(256), 0, 0, A, N);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) (void)time_one<256, 8, 0, 0>(1024);
    printf("ms per GiB each way: one-shot grid / persistent grid of 1024 threads per CU\n");
    row<64, 1>(); row<64, 2>(); row<64, 4>(); row<64, 8>(); row<64, 16>(); row<64, 32>();
    row<256, 1>(); row<256, 2>(); row<256, 4>(); row<256, 8>();
    row<512, 1>(); row<512, 2>(); row<512, 4>(); row<512, 8>();
    row<1024, 1>(); row<1024, 2>(); row<1024, 4>();
    row<256, 8>();
    return 0;
}
