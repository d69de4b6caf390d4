// fir_lat.hip -- small-call direct-form FIR arranged for latency (design notes: fir_lat.hip.h).
#include "fir_lat.hip.h"

namespace qk {

namespace {
typedef float v2f_t __attribute__((ext_vector_type(2)));
}

__global__ __launch_bounds__(256) void fir_lat_kernel(const FirLatArgs a) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int N = a.N, H = N - 1, Np = a.Np;
    const int nwg = (a.nwaves + 3) / 4;
    if ((int)blockIdx.x == nwg) {
        // history for the next call: the last N - 1 samples of hist ++ in
        for (int i = t; i < H; i += 256) {
            const long long g = a.count - H + i;
            a.hist_next[i] = g < 0 ? a.hist[g + H] : a.in[g];
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char fl_smem[];
    float* tapl = reinterpret_cast<float*>(fl_smem);                        // h[0..N), zeros up to Np + 8
    const int wlen = 64 + Np + 8;                                           // samples a wave stages (the tail beyond 64 + N - 1 meets zero taps)
    v2f_t* win = reinterpret_cast<v2f_t*>(tapl + Np + 8) + (size_t)w * wlen;
    for (int j = t; j < Np + 8; j += 256) tapl[j] = j < N ? a.taps[j] : 0.0f;
    const int tile = (int)blockIdx.x * 4 + w;
    const long long j0 = 64LL * tile - H;                                   // stream position (relative to in[0]) of the window's slot 0
    if (tile < a.nwaves) {
        for (int x = l; x < wlen; x += 64) {
            const long long g = j0 + x;
            float2 v = make_float2(0.0f, 0.0f);
            if (g < 0) { if (g + H >= 0) v = a.hist[g + H]; }
            else if (g < a.count) v = a.in[g];
            win[x] = (v2f_t){v.x, v.y};
        }
    }
    __syncthreads();                                                        // (the taps; the window is the wave's own)
    if (tile >= a.nwaves) return;
    const v2f_t* xp = win + l;                                              // x[l + k]
    const float4* hp = reinterpret_cast<const float4*>(tapl);
    v2f_t acc = {0.0f, 0.0f};
    v2f_t xv[8], xn[8];
    float4 h0 = hp[0], h1 = hp[1], g0, g1;
#pragma unroll
    for (int k = 0; k < 8; k++) xv[k] = xp[k];
    const int nfull = N & ~7;
    for (int c = 0; c < nfull; c += 8) {
        // (the reads one chunk ahead stay inside the arrays: tapl and win are 8 entries longer than the loop needs)
        g0 = hp[(c >> 2) + 2];
        g1 = hp[(c >> 2) + 3];
#pragma unroll
        for (int k = 0; k < 8; k++) xn[k] = xp[c + 8 + k];
        acc = __builtin_elementwise_fma((v2f_t){h0.x, h0.x}, xv[0], acc);
        acc = __builtin_elementwise_fma((v2f_t){h0.y, h0.y}, xv[1], acc);
        acc = __builtin_elementwise_fma((v2f_t){h0.z, h0.z}, xv[2], acc);
        acc = __builtin_elementwise_fma((v2f_t){h0.w, h0.w}, xv[3], acc);
        acc = __builtin_elementwise_fma((v2f_t){h1.x, h1.x}, xv[4], acc);
        acc = __builtin_elementwise_fma((v2f_t){h1.y, h1.y}, xv[5], acc);
        acc = __builtin_elementwise_fma((v2f_t){h1.z, h1.z}, xv[6], acc);
        acc = __builtin_elementwise_fma((v2f_t){h1.w, h1.w}, xv[7], acc);
        h0 = g0;
        h1 = g1;
#pragma unroll
        for (int k = 0; k < 8; k++) xv[k] = xn[k];
    }
    // the last N % 8 taps: no padding taps are applied (0 * Inf would poison outputs whose window does not hold the sample)
    {
        const float hk[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (nfull + k < N) acc = __builtin_elementwise_fma((v2f_t){hk[k], hk[k]}, xv[k], acc);
    }
    const long long o = 64LL * tile + l;
    if (o < a.count) a.out[o] = make_float2(acc.x, acc.y);
}

int launch_fir_lat(const FirLatArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(fir_lat_kernel, dim3((a.nwaves + 3) / 4 + 1), dim3(256), fir_lat_lds_bytes(a.Np), stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
