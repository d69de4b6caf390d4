// burst_policy.hip -- third skeleton run: cache-policy bits of the loads and stores of a "load segment, store segment"
// kernel (1 GiB in, 1 GiB out; a 256-thread workgroup per PER x 4 KB segment, rows of 4 KB).  Loads and stores go through
// buffer instructions so that the aux field can be set: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1, 18 sc1 nt, 3 sc0 nt.
//   hipcc --offload-arch=gfx950 -O3 -o burst_policy burst_policy.hip && ./burst_policy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int PER, int LP, int SP>
__global__ __launch_bounds__(256) void seg(const f4* __restrict__ in, f4* __restrict__ out, int nseg) {
    const int t = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 30, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 1 << 30, 0x00020000);
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const unsigned base = (unsigned)s * (PER * 4096u) + t * 16u;
        f4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, base + k * 4096u, 0, LP));
#pragma unroll
        for (int k = 0; k < PER; k++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v[k]), rs_out, base + k * 4096u, 0, SP);
    }
}

__global__ void fill(f4* a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = (f4){1.0f, 0.5f, 0.25f, 0.125f};
}

static f4 *A, *B;
static const long long N = 1ll << 26;
static hipEvent_t e0, e1;

template <int PER, int LP, int SP> float time_one(int grid) {
    const int nseg = (int)(N / (PER * 256));
    const int g = grid ? grid : nseg;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((seg<PER, LP, SP>), dim3(g), dim3(256), 0, 0, A, B, nseg);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((seg<PER, LP, SP>), dim3(g), dim3(256), 0, 0, A, B, nseg);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int PER, int LP> void row() {
    printf("PER %d load aux %2d | store aux 0: %.3f / %.3f   1: %.3f / %.3f   2: %.3f / %.3f   3: %.3f / %.3f   16: %.3f / %.3f   17: %.3f / %.3f   18: %.3f / %.3f\n", PER, LP,
           time_one<PER, LP, 0>(0), time_one<PER, LP, 0>(1024), time_one<PER, LP, 1>(0), time_one<PER, LP, 1>(1024), time_one<PER, LP, 2>(0),
           time_one<PER, LP, 2>(1024), time_one<PER, LP, 3>(0), time_one<PER, LP, 3>(1024), time_one<PER, LP, 16>(0), time_one<PER, LP, 16>(1024),
           time_one<PER, LP, 17>(0), time_one<PER, LP, 17>(1024), time_one<PER, LP, 18>(0), time_one<PER, LP, 18>(1024));
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&A, N * 16);
    (void)hipMalloc(&B, N * 16);
    hipLaunchKernelGGL(fill, dim3((unsigned)(N / 256)), dim3(256), 0, 0, A, N);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) (void)time_one<8, 0, 0>(1024);
    printf("ms per GiB each way: one-shot grid / persistent grid of 1024\n");
    row<8, 0>(); row<8, 1>(); row<8, 2>(); row<8, 3>(); row<8, 16>(); row<8, 17>(); row<8, 18>();
    row<4, 0>(); row<4, 2>(); row<4, 18>();
    row<1, 0>(); row<1, 2>();
    row<8, 0>();
    return 0;
}
