// burst_map.hip -- follow-up to burst_copy.hip: a 256-thread workgroup copies one 32 KB segment (1 GiB in, 1 GiB out);
// what differs between the rows is WHICH 16 bytes each lane's k-th load / store touches.
//   map 0  "rows of 4 KB":   load k of the workgroup is one contiguous 4 KB row; a wave's eight loads are 4 KB apart
//   map 1  "wave-contiguous": a wave owns 8 KB; its k-th load is the k-th KB of it
//   map 2  "fir_fft":         the overlap-save kernel's pattern -- 2 KB rows (256 samples), lanes 0-31 take 512 B of row k,
//                             lanes 32-63 the same 512 B of row k + 8
//   map 3  "fir_fft 8 B":     one sample (8 B) per lane: 512 B per wave and row, sixteen rows
//   map 4  "lane-contiguous": a lane owns 128 contiguous bytes (eight 16 B loads 16 B apart)
// side: 0 copy, 1 read only, 2 write only.  nt: non-temporal loads and stores.
//   hipcc --offload-arch=gfx950 -O3 -o burst_map burst_map.hip && ./burst_map
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MAP> __device__ __forceinline__ long long off16(int t, int k) {   // offset in f4 units inside the 32 KB segment
    const int w = t >> 6, l = t & 63;
    if (MAP == 0) return k * 256 + t;
    if (MAP == 1) return w * 512 + k * 64 + l;
    if (MAP == 2) return (long long)(k + 8 * (l >> 5)) * 128 + w * 32 + (l & 31);
    if (MAP == 4) return t * 8 + k;
    return 0;
}

template <int MAP, int SIDE, bool NT>
__global__ __launch_bounds__(256) void seg(const f4* __restrict__ in, f4* __restrict__ out, int nseg) {
    const int t = threadIdx.x;
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const f4* p = in + (long long)s * 2048;
        f4* q = out + (long long)s * 2048;
        if (MAP == 3) {
            const f2* p2 = reinterpret_cast<const f2*>(p);
            f2* q2 = reinterpret_cast<f2*>(q);
            f2 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = SIDE == 2 ? (f2){1.f, 2.f} : (NT ? __builtin_nontemporal_load(p2 + k * 256 + t) : p2[k * 256 + t]);
            if (SIDE == 1) {
                float a = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) a += v[k].x + v[k].y;
                if (a == 12345.f) q2[t] = v[0];
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (NT) __builtin_nontemporal_store(v[k], q2 + k * 256 + t);
                    else q2[k * 256 + t] = v[k];
                }
            }
        } else {
            f4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                v[k] = SIDE == 2 ? (f4){1.f, 2.f, 3.f, 4.f} : (NT ? __builtin_nontemporal_load(p + off16<MAP>(t, k)) : p[off16<MAP>(t, k)]);
            if (SIDE == 1) {
                float a = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) a += v[k].x + v[k].y + v[k].z + v[k].w;
                if (a == 12345.f) q[t] = v[0];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (NT) __builtin_nontemporal_store(v[k], q + off16<MAP>(t, k));
                    else q[off16<MAP>(t, k)] = v[k];
                }
            }
        }
    }
}

__global__ void fill(f4* a, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = (f4){1.0f, 0.5f, 0.25f, 0.125f};
}

static f4 *A, *B;
static const long long N = 1ll << 26;
static hipEvent_t e0, e1;

template <int MAP, int SIDE, bool NT> float time_one(int grid) {
    const int nseg = (int)(N / 2048);
    const int g = grid ? grid : nseg;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((seg<MAP, SIDE, NT>), dim3(g), dim3(256), 0, 0, A, B, nseg);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((seg<MAP, SIDE, NT>), dim3(g), dim3(256), 0, 0, A, B, nseg);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int MAP> void row(const char* name) {
    printf("%-16s copy: one-shot %.3f  grid 1024 %.3f  grid 4096 %.3f | nt %.3f / %.3f | read only %.3f / %.3f | write only %.3f / %.3f (nt %.3f)\n", name,
           time_one<MAP, 0, false>(0), time_one<MAP, 0, false>(1024), time_one<MAP, 0, false>(4096), time_one<MAP, 0, true>(0),
           time_one<MAP, 0, true>(1024), time_one<MAP, 1, false>(0), time_one<MAP, 1, false>(1024), time_one<MAP, 2, false>(0),
           time_one<MAP, 2, false>(1024), time_one<MAP, 2, true>(0));
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&A, N * 16);
    (void)hipMalloc(&B, N * 16);
    hipLaunchKernelGGL(fill, dim3((unsigned)(N / 256)), dim3(256), 0, 0, A, N);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) (void)time_one<0, 0, false>(1024);
    printf("ms per GiB each way (copy) / per GiB (read only, write only); 32 KB segment per 256-thread workgroup\n");
    row<0>("rows of 4 KB");
    row<1>("wave-contiguous");
    row<2>("fir_fft 16 B");
    row<3>("fir_fft 8 B");
    row<4>("lane-contiguous");
    row<0>("rows of 4 KB");
    return 0;
}
