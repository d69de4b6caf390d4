// fft_fir.hip.h -- overlap-save fast convolution for long complex FIRs (gfx950).
//
// Why: direct form costs 2*ntaps FMAs per complex sample (1024 FLOP at 256 taps against 16
// algorithmic bytes: FP32-compute-bound at ~30 % of the HBM roofline).  A
// 4096-point overlap-save block costs ~135 FLOP per sample, which puts the filter back
// under the HBM roof.  Same operator, same state, same boundary (FIR<complex_t>::run,
// src/dsp/filter.h:51-74); results differ from the k-ordered FP32 sum only by FP32 FFT
// rounding (~3e-7 RMS relative, tests/test_gpu_parity.py) -- inside the 1e-5 bar.
//
// One workgroup = 256 lanes = one 4096-point block at a time, persistent over blocks.
//   x (F samples, H = ntaps-1 of them overlap) --A--> --B--> --C--> * Hf --C'--> --B'--> --A'--> y
// F = 16*16*16: three radix-16 passes done in registers (16 points per lane), separated by
// LDS transposes; the inverse runs the same passes backwards, so no bit-reversal pass
// exists: Hf is stored in the digit-reversed order pass C leaves the spectrum in.
// Per-lane constants (pass twiddles, Hf slice) are loaded once per workgroup.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int kFftN = 4096;
constexpr int kFftNT = 256;
constexpr int kFftRow1 = 272;   // layout 1: [k0][n1*16 + n0], row padded 256 -> 272 elements
constexpr int kFftRow2 = 17;    // layout 2: [k0*16 + k1][n0], row padded 16 -> 17 elements
constexpr int kFftLdsElems = 16 * kFftRow1;  // 4352 >= 256*17 = 4352

struct FftArgs {
    const float2* in;
    float2* out;
    const float2* hist;       // H samples
    float2* hist_next;
    const float2* Hf;         // [256][16]: Hf[(k0*16+k1)*16 + k2] = FFT(taps reversed)[k0 + 16 k1 + 256 k2] / F
    const float2* TA;         // [256][16]: exp(-j 2pi t k / 4096)
    const float2* TB;         // [16][16] : exp(-j 2pi lo k / 256)
    long long count;
    int H;                    // ntaps - 1
    int L;                    // valid outputs per block = F - H
    int nblocks;              // ceil(count / L)
    int nwg;                  // persistent workgroups (grid = nwg + 1; the last one hands over history)
};

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
__global__ __launch_bounds__(kFftNT, 2) void fir_fft_kernel(const FftArgs a) {
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
            for (int n2 = 0; n2 < 16; n2++) v[n2] = p[n2 * 256];
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
                if (n2 * 256 + t >= H) a.out[o0 + n2 * 256] = v[rev16(n2)];
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const long long n = o0 + n2 * 256;
                if (n2 * 256 + t >= H && n < a.count) a.out[n] = v[rev16(n2)];
            }
        }
    }
}

}  // namespace qk
