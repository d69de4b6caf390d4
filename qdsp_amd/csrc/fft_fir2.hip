// fft_fir2.hip -- overlap-save FIR, two segments per workgroup (the production FIR<complex_t>
// kernel for large calls; fft_fir.hip keeps the one-segment form used by the decimators).
//
// Same algorithm, layouts and tables as fir_fft_kernel<1> (fft_fir.hip.h / fft_fir.hip), with one
// change of data layout in registers: a workgroup transforms segments 2u and 2u+1 TOGETHER and
// every value is held as a 2-vector (segment 2u, segment 2u+1) -- real parts in one aligned
// register pair, imaginary parts in another.  All butterfly / twiddle / spectrum arithmetic is
// then elementwise on 2-vectors with the twiddle broadcast to both halves, which hipcc lowers
// to v_pk_add/mul/fma_f32 by itself (no inline asm, no shuffles: the two halves never mix).
// gfx950 issues a packed FP32 op in the time of a scalar one (PMC: one quad-cycle per VALU
// instruction either way), so the VALU work per segment halves: the scalar kernel was
// VALU-bound (78 % VALU busy at 1329 instructions per segment and wave).
// LDS: each transpose buffer becomes two planes (re pairs, im pairs) with the proven
// conflict-free element layouts; 74 KB per workgroup -> 2 workgroups (8 waves) per CU, each
// wave carrying two segments: the same number of segments in flight as before.
#include "fft_fir.hip.h"

namespace qk {

typedef float f2 __attribute__((ext_vector_type(2)));
struct C2 {   // two complex numbers (one per segment), split re / im
    f2 x, y;
};
__device__ __forceinline__ f2 bc(float s) { f2 r; r.x = s; r.y = s; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return C2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return C2{a.x - b.x, a.y - b.y}; }
// (both segments) * w, or * conj(w); w is one complex scalar
template <bool CONJ> __device__ __forceinline__ C2 cmulc(C2 a, float2 w) {
    const f2 wr = bc(w.x), wi = bc(w.y);
    if (CONJ) return C2{fma2(a.x, wr, a.y * wi), fma2(a.y, wr, -(a.x * wi))};
    return C2{fma2(a.x, wr, -(a.y * wi)), fma2(a.x, wi, a.y * wr)};
}
// multiply by -j (forward) / +j (inverse): a register rename plus one negation
template <bool INV> __device__ __forceinline__ C2 mulj(C2 a) { return INV ? C2{-a.y, a.x} : C2{a.y, -a.x}; }

template <bool INV> __device__ __forceinline__ void fft4(C2& a0, C2& a1, C2& a2, C2& a3) {
    const C2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mulj<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

__host__ __device__ constexpr int rev16b(int k) { return 4 * (k & 3) + (k >> 2); }
__host__ __device__ constexpr int pos1b(int e) { return (e >> 1) + 136 * (e & 1); }

// In-register 16-point DFT (radix 4x4) of both segments; X[k] is left at v[rev16b(k)].
template <bool INV> __device__ __forceinline__ void fft16(C2 (&v)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
#pragma unroll
    for (int n0 = 0; n0 < 4; n0++) fft4<INV>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
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

// FIR<complex_t>::run (src/dsp/filter.h:51-74) by overlap-save, segments 2u and 2u+1 per pass.
__global__ __launch_bounds__(kFftNT, 2) void fir_fft2_kernel(const FftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2* lre = reinterpret_cast<f2*>(smem);              // plane of (seg A, seg B) real parts
    f2* lim = lre + kFftLdsElems;                       // plane of imaginary parts
    float2* tbl = reinterpret_cast<float2*>(lim + kFftLdsElems);   // pass-B twiddles, rows padded to 17
    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int H = a.H;
    const int te = (t & ~63) | ((t & 31) << 1) | ((t >> 5) & 1);   // see fft_fir.hip: 16-byte I/O by lane pairs
    const int half = (t >> 5) & 1;
    const int pte = pos1b(te);

    if ((int)blockIdx.x == a.nwg) {
        for (int i = t; i < H; i += kFftNT) {   // history hand-over (filter.h:71)
            const long long g = a.count - H + i;
            a.hist_next[i] = (g < 0) ? a.hist[g + H] : a.in[g];
        }
        return;
    }

    float2 ta[16], hf[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hf[k] = a.Hf[t * 16 + k];
        ta[k] = a.TA[te * 16 + k];
    }
    tbl[(t >> 4) * 17 + (t & 15)] = a.TB[t];
    const float2* tb = tbl + lo * 17;

    const int npairs = (a.nblocks + 1) >> 1;
    for (int u = blockIdx.x; u < npairs; u += a.nwg) {
        C2 v[16];
        // ---- load both segments; lane ends up with elements n2*256 + te of each ---------------
#pragma unroll
        for (int sgm = 0; sgm < 2; sgm++) {
            const int b = 2 * u + sgm;
            const long long seg0 = (long long)b * a.L - a.seg_shift;
            const bool live = b < a.nblocks;
            const bool interior = live && seg0 >= 0 && seg0 + kFftN <= a.count;
            float2 w[16];
            if (interior && a.vec) {
                const float4* __restrict__ p4 = reinterpret_cast<const float4*>(a.in + seg0 + (te & ~1)) + half * 8 * 128;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float4 q4 = p4[r * 128];
                    const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.x), __float_as_uint(q4.z), false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.y), __float_as_uint(q4.w), false, false);
                    w[r] = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
                    w[8 + r] = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
                }
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const long long g = seg0 + n2 * 256 + te;
                    float2 x = make_float2(0.0f, 0.0f);
                    if (live) {
                        if (g < 0) { if (g + H >= 0) x = a.hist[g + H]; }
                        else if (g < a.count) x = a.in[g];
                    }
                    w[n2] = x;
                }
            }
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                if (sgm == 0) { v[n2].x.x = w[n2].x; v[n2].y.x = w[n2].y; }
                else { v[n2].x.y = w[n2].x; v[n2].y.y = w[n2].y; }
            }
        }
        // ---- pass A (over n2) + twiddle W4096^(te*k0) -----------------------------------------
        fft16<false>(v);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const C2 o = (k == 0) ? v[rev16b(0)] : cmulc<false>(v[rev16b(k)], ta[k]);
            lre[k * kFftRow1 + pte] = o.x;
            lim[k * kFftRow1 + pte] = o.y;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            v[j].x = lre[hi * kFftRow1 + pos1b(j * 16 + lo)];
            v[j].y = lim[hi * kFftRow1 + pos1b(j * 16 + lo)];
        }
        // ---- pass B (over n1) + twiddle W256^(n0*k1) ------------------------------------------
        fft16<false>(v);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const C2 o = (k == 0) ? v[rev16b(0)] : cmulc<false>(v[rev16b(k)], tb[k]);
            lre[(hi * 16 + k) * kFftRow2 + lo] = o.x;
            lim[(hi * 16 + k) * kFftRow2 + lo] = o.y;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            v[j].x = lre[t * kFftRow2 + j];
            v[j].y = lim[t * kFftRow2 + j];
        }
        // ---- pass C (over n0), spectrum * Hf, pass C' (over k2) ---------------------------------
        fft16<false>(v);
        C2 y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cmulc<false>(v[rev16b(k)], hf[k]);
        fft16<true>(y);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const C2 o = (j == 0) ? y[rev16b(0)] : cmulc<true>(y[rev16b(j)], tb[j]);
            lre[t * kFftRow2 + j] = o.x;
            lim[t * kFftRow2 + j] = o.y;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            v[j].x = lre[(hi * 16 + j) * kFftRow2 + lo];
            v[j].y = lim[(hi * 16 + j) * kFftRow2 + lo];
        }
        // ---- pass B' (over k1) ----------------------------------------------------------------
        fft16<true>(v);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            lre[hi * kFftRow1 + pos1b(j * 16 + lo)] = v[rev16b(j)].x;
            lim[hi * kFftRow1 + pos1b(j * 16 + lo)] = v[rev16b(j)].y;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const C2 e = C2{lre[k * kFftRow1 + pte], lim[k * kFftRow1 + pte]};
            v[k] = (k == 0) ? e : cmulc<true>(e, ta[k]);
        }
        // ---- pass A' (over k0) and store the valid outputs of both segments -------------------
        fft16<true>(v);
#pragma unroll
        for (int sgm = 0; sgm < 2; sgm++) {
            const int b = 2 * u + sgm;
            const long long seg0 = (long long)b * a.L - a.seg_shift;
            const bool live = b < a.nblocks;
            const bool interior = live && seg0 >= 0 && seg0 + kFftN <= a.count;
            float2 w[16];
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const C2 o = v[rev16b(n2)];
                w[n2] = (sgm == 0) ? make_float2(o.x.x, o.y.x) : make_float2(o.x.y, o.y.y);
            }
            if (interior && a.vec) {
                float4* __restrict__ o4 = reinterpret_cast<float4*>(a.out + seg0 + (te & ~1)) + half * 8 * 128;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[r].x), __float_as_uint(w[8 + r].x), false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[r].y), __float_as_uint(w[8 + r].y), false, false);
                    if ((half * 8 + r) * 256 + (te & ~1) >= a.ov)   // ov is even: both elements or neither
                        o4[r * 128] = make_float4(__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sx[1]), __uint_as_float(sy[1]));
                }
            } else if (live) {
                const long long o0 = seg0 + te;
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const long long n = o0 + n2 * 256;
                    if (n2 * 256 + te >= a.ov && n < a.nout) a.out[n] = w[n2];
                }
            }
        }
    }
}

int launch_fir_fft2(const FftArgs& a, int grid, hipStream_t stream) {
    const size_t lds = 2 * (size_t)kFftLdsElems * sizeof(f2) + 16 * 17 * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fir_fft2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(fir_fft2_kernel, dim3(grid), dim3(kFftNT), lds, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
