// fft_fir.hip -- the overlap-save FIR kernel (design notes: fft_fir.hip.h).
//
// Own translation unit because it is compiled with -fno-slp-vectorize: left on, clang's SLP
// pass packs the butterflies' re/im arithmetic into v_pk_add/mul/fma_f32 and pays for it
// with ~450 v_mov/v_xor shuffles per 4096-point block (multiply-by-j swaps) and 209 VGPRs
// (2 waves/SIMD); scalar f32 VALU ops issue at the same FLOP rate on gfx950's 32-wide
// SIMDs once two waves share a SIMD, need no shuffles, and fit in 126 VGPRs (4 waves/SIMD).
#include "fft_fir.hip.h"

namespace qk {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * b, and a * conj(b)
template <bool CONJ> __device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    if (CONJ) return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// multiply by -j (forward) / +j (inverse)
template <bool INV> __device__ __forceinline__ float2 mulj(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <bool INV>
__device__ __forceinline__ void fft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mulj<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

// In-register 16-point DFT, radix 4x4.  Input v[n]; output X[k] is left at v[rev16(k)].
__host__ __device__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }

template <bool INV> __device__ __forceinline__ void fft16(float2 (&v)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
#pragma unroll
    for (int n0 = 0; n0 < 4; n0++) fft4<INV>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    // v[4*k0 + n0] *= W16^(n0*k0); forward W = exp(-j 2pi/16), inverse conj
    const float2 w1 = make_float2(c1, -s1), w2 = make_float2(r, -r), w3 = make_float2(s1, -c1);
    const float2 w6 = make_float2(-r, -r), w9 = make_float2(-c1, s1);
    v[4 * 1 + 1] = cmulc<INV>(v[4 * 1 + 1], w1);
    v[4 * 1 + 2] = cmulc<INV>(v[4 * 1 + 2], w2);
    v[4 * 1 + 3] = cmulc<INV>(v[4 * 1 + 3], w3);
    v[4 * 2 + 1] = cmulc<INV>(v[4 * 2 + 1], w2);
    v[4 * 2 + 2] = mulj<INV>(v[4 * 2 + 2]);
    v[4 * 2 + 3] = cmulc<INV>(v[4 * 2 + 3], w6);
    v[4 * 3 + 1] = cmulc<INV>(v[4 * 3 + 1], w3);
    v[4 * 3 + 2] = cmulc<INV>(v[4 * 3 + 2], w6);
    v[4 * 3 + 3] = cmulc<INV>(v[4 * 3 + 3], w9);
#pragma unroll
    for (int k0 = 0; k0 < 4; k0++) fft4<INV>(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}

// y[n] = sum_k taps[k] * s[n - H + k], s = hist ++ in, by overlap-save (see file header).
// NT: stream the samples with non-temporal loads/stores (each is touched once; keeps the
// tables and the 255-sample overlap, not the stream, in L2 / Infinity Cache).
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float2 ld_stream(const float2* p) {
    if (NT) {
        const v2f_t r = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(p));
        return make_float2(r.x, r.y);
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st_stream(float2* p, float2 v) {
    if (NT) {
        v2f_t r;
        r.x = v.x;
        r.y = v.y;
        __builtin_nontemporal_store(r, reinterpret_cast<v2f_t*>(p));
    } else {
        *p = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(kFftNT, 4) void fir_fft_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[kFftLdsElems + 16 * 17];
    float2* tbl = lds + kFftLdsElems;  // pass-B twiddles W256^(lo*k), rows padded to 17
    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int H = a.H;

    if ((int)blockIdx.x == a.nwg) {
        // history hand-over (filter.h:71): last H samples of hist ++ in -> the other buffer
        for (int i = t; i < H; i += kFftNT) {
            const long long g = a.count - H + i;
            a.hist_next[i] = (g < 0) ? a.hist[g + H] : a.in[g];
        }
        return;
    }

    // per-lane constants, loaded once
    float2 ta[16], hf[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        ta[k] = a.TA[t * 16 + k];
        hf[k] = a.Hf[t * 16 + k];
    }
    tbl[(t >> 4) * 17 + (t & 15)] = a.TB[t];
    const float2* tb = tbl + lo * 17;

    for (int b = blockIdx.x; b < a.nblocks; b += a.nwg) {
        const long long seg0 = (long long)b * a.L - H;  // stream index of segment element 0
        float2 v[16];
        // ---- load: lane t takes elements n2*256 + t ------------------------------------
        if (seg0 >= 0 && seg0 + kFftN <= a.count) {
            const float2* __restrict__ p = a.in + seg0 + t;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) v[n2] = ld_stream<NT>(p + n2 * 256);
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const long long g = seg0 + n2 * 256 + t;
                v[n2] = (g < 0) ? a.hist[g + H] : (g < a.count ? a.in[g] : make_float2(0.0f, 0.0f));
            }
        }
        // ---- pass A (over n2) + twiddle W4096^(t*k0) -----------------------------------
        fft16<false>(v);
        __syncthreads();  // previous block's last LDS reads are done
#pragma unroll
        for (int k = 0; k < 16; k++) lds[k * kFftRow1 + t] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], ta[k]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[hi * kFftRow1 + j * 16 + lo];
        // ---- pass B (over n1) + twiddle W256^(n0*k1) -----------------------------------
        fft16<false>(v);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[(hi * 16 + k) * kFftRow2 + lo] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], tb[k]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[t * kFftRow2 + j];
        // ---- pass C (over n0), spectrum * Hf, pass C' (over k2) ----------------------------
        fft16<false>(v);
        {
            float2 y[16];
#pragma unroll
            for (int k = 0; k < 16; k++) y[k] = cmulc<false>(v[rev16(k)], hf[k]);
            fft16<true>(y);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++)
                lds[t * kFftRow2 + j] = (j == 0) ? y[rev16(0)] : cmulc<true>(y[rev16(j)], tb[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[(hi * 16 + j) * kFftRow2 + lo];
        // ---- pass B' (over k1) -------------------------------------------------------------
        fft16<true>(v);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) lds[hi * kFftRow1 + j * 16 + lo] = v[rev16(j)];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float2 e = lds[k * kFftRow1 + t];
            v[k] = (k == 0) ? e : cmulc<true>(e, ta[k]);
        }
        // ---- pass A' (over k0) and store the L valid outputs ------------------------------
        fft16<true>(v);
        const long long o0 = (long long)b * a.L - H + t;  // output index of element t (n2 = 0)
        if (seg0 >= 0 && seg0 + kFftN <= a.count) {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                if (n2 * 256 + t >= H) st_stream<NT>(a.out + o0 + n2 * 256, v[rev16(n2)]);
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const long long n = o0 + n2 * 256;
                if (n2 * 256 + t >= H && n < a.count) a.out[n] = v[rev16(n2)];
            }
        }
    }
}


int launch_fir_fft(const FftArgs& a, int grid, hipStream_t stream) {
    if (a.nt) hipLaunchKernelGGL(fir_fft_kernel<true>, dim3(grid), dim3(kFftNT), 0, stream, a);
    else hipLaunchKernelGGL(fir_fft_kernel<false>, dim3(grid), dim3(kFftNT), 0, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
