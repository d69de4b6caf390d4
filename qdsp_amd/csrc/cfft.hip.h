// cfft.hip.h -- scalar complex helpers and the in-register radix-4x4 16-point DFT shared by the
// overlap-save FIR kernels (fft_fir.hip) and the polyphase channelizer (chan.hip).  Both
// translation units are built with -fno-slp-vectorize (see fft_fir.hip).
#pragma once
#include <hip/hip_runtime.h>

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

// 64-bit fixed-point phase (2^64 == one turn) -> unit phasor, FP64
__device__ __forceinline__ double2 fx_phasor(unsigned long long ph) {
    const double t = (double)(ph >> 11) * (1.0 / 9007199254740992.0);
    double s, c;
    sincospi(2.0 * t, &s, &c);
    return make_double2(c, s);
}
__device__ __forceinline__ double2 dcmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

}  // namespace qk
