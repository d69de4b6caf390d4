// fft1k_fir.hip -- overlap-save with 1024-point segments, one WAVE per segment: the form for reference-sized
// calls (the block API hands over at most 1e6 samples per call, src/dsp/stream.h:7).
//
// Why a second overlap-save kernel: fir_fft_kernel's 4096-point segment is one 256-lane workgroup with eight
// barriers on its critical path; a lone workgroup takes ~12 us from first load to last store whether the CU is
// shared or not (measured: a 1e6-sample call = 245 segments on 256 CUs = 12.0 us, and 2^27 samples = 34 rounds of
// 4 workgroups per CU = 0.41 ms), so every call below ~4e6 samples costs those 12 us.  Here a segment is 1024
// points held by ONE wave (16 per lane): no workgroup barrier at all -- the four transposes go through a
// wave-private 10.5 KB LDS tile and are ordered by the LDS queue itself -- and a 1e6-sample call spreads over
// 1300 waves (every SIMD of the chip) instead of 245 workgroups.  It pays with overlap: at 256 taps 769 of 1024
// points are new (94 % of 4096), so the chip-filling calls stay with the 4096-point kernels.
//
//   1024 = 16 x 16 x 4.  n = 64 i + l (l = lane), k = ka + 16 (kb1 + 16 kb0):
//   pass A   radix 16 over i (registers), twiddle W1024^(l ka)                 lane l          holds ka  = 0..15
//   LDS      [ka][l]  ->  lane (ka, j = l & 3) reads l = j + 4 m
//   pass B   radix 16 over m, twiddle W64^(j kb1)                              lane (ka, j)    holds kb1 = 0..15
//   LDS      [ka][g + 5 j + 20 s], kb1 = 4 g + s  ->  lane (ka, g) reads its 4 values of s x 4 values of j
//   pass C   4 x radix 4 over j; spectrum product; pass C' back                lane (ka, g)    holds (s, kb0)
//   ... and the same road back (conjugate twiddles), so no bit-reversal pass exists: Hf is stored in the order
//   pass C leaves the spectrum in.  Row pitch 84 (= 4 mod 16) makes all four access patterns conflict-free for
//   8-byte LDS accesses (16 lanes per phase, 32 banks).
// Tried for the chip-filling calls and rejected (2^27 samples, 256 taps, same box and run): the same segments as
// persistent workgroups of 12 waves (3 per SIMD), the three tables in LDS once per workgroup, the next segment's loads
// in flight behind the current transform, each XCD on a contiguous eighth of every round of segments so that the
// overlap is re-read from its L2 (FETCH_SIZE x 1.09 of the algorithmic bytes: it is): 535-545 us against 464-470 us for
// fir_fft_kernel<1>.  Counters: 688 VALU instructions per segment (0.63x the 4096-point kernel's per output) = 223 us of
// issue, LDS busy 253 us per CU (88 KB of transposes + table reads per segment), 2.26 GB of HBM traffic, waves waiting
// 44 % of their cycles -- three half-used resources that three waves per SIMD do not overlap; with the tables in
// registers instead (2 waves per SIMD, no prefetch) the same 535 us.  At 63 taps (94 % new points) it reaches 455 us
// against 464: not worth a second kernel.
// Same operator, state and semantics as fir_fft_kernel<1, ROT> (FIR, any-decimation resampler through the
// strided store, fused VFO); built with -fno-slp-vectorize like the other FFT translation units.
#include "fft_fir.hip.h"
#include "cfft.hip.h"
#include "cpk.hip.h"

namespace qk {

constexpr int kF1P = kFft1kPitch;

// The LDS tile is private to the wave and the LDS queue runs a wave's accesses in order; what is left to order is
// the compiler: lanes read what OTHER lanes wrote, which it cannot see.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64-bit fixed-point phase -> unit phasor in FP32: the quarter turn is taken off exactly in integers, the rest
// (|x| <= pi/4, 2^-34 turn resolution) goes through Taylor polynomials whose first dropped terms are 2e-9 / 1e-10.
// ~1e-7 absolute, against 3e-8 for the FP64 sincospi of fx_phasor -- and ~25 instructions against ~400, which
// matters where the phasor sits on the critical path of a latency-bound call.
__device__ __forceinline__ float2 fx_phasor_f32(unsigned long long ph) {
    const unsigned long long qt = (ph + (1ULL << 61)) >> 62;          // nearest quarter turn, mod 4
    const long long r = (long long)(ph - (qt << 62));                 // [-2^61, 2^61): the remainder, 2^64 = one turn
    const float x = (float)(int)(r >> 30) * 3.6572952e-10f;           // 2 pi / 2^34
    const float x2 = x * x;
    float sn = fmaf(x2, 2.7557319e-6f, -1.9841270e-4f);
    sn = fmaf(x2, sn, 8.3333333e-3f);
    sn = fmaf(x2, sn, -1.6666667e-1f);
    sn = fmaf(x * x2, sn, x);
    float cs = fmaf(x2, -2.7557319e-7f, 2.4801587e-5f);
    cs = fmaf(x2, cs, -1.3888889e-3f);
    cs = fmaf(x2, cs, 4.1666667e-2f);
    cs = fmaf(x2, cs, -0.5f);
    cs = fmaf(x2, cs, 1.0f);
    const int qi = (int)qt;
    const float c = (qi & 1) ? -sn : cs, s2 = (qi & 1) ? cs : sn;      // + quarter turn: (c, s) -> (-s, c)
    return (qi & 2) ? make_float2(-c, -s2) : make_float2(c, s2);
}

// ---- the pieces both kernels are made of -------------------------------------------------------------------

// history hand-over (filter.h:71 / resampling.h:129): element i of the last H samples of hist ++ in
template <bool ROT>
__device__ __forceinline__ void f1k_hand_over(const FftArgs& a, int i) {
    const int H = a.H;
    if (i >= H) return;
    const long long g = a.count - H + i;
    float2 v;
    if (g < 0) {
        v = a.hist_keep[g + H];
        if (ROT && a.hist_raw_next) a.hist_raw_next[i] = a.hist[g + H];
    } else {
        v = a.in[g];
        if (ROT) {
            const double2 p = fx_phasor(a.phase_in0 + (unsigned long long)g * a.dphase);
            const float gain = fmaf((float)(int)(g & 511), a.gm1, 1.0f);
            if (a.hist_raw_next) a.hist_raw_next[i] = make_float2(v.x * gain, v.y * gain);
            v = cmulc<false>(v, make_float2((float)p.x * gain, (float)p.y * gain));
        }
    }
    a.hist_next[i] = v;
}

// the lane's 16 elements 64 i + l of the segment whose element 0 is stream position seg0
__device__ __forceinline__ void f1k_load(const FftArgs& a, long long seg0, int l, v2f (&v)[16]) {
    if (seg0 >= 0 && seg0 + kFft1kN <= a.count) {
        const float2* __restrict__ p = a.in + seg0 + l;
#pragma unroll
        for (int i = 0; i < 16; i++) { const float2 x = p[64 * i]; v[i] = mk2(x.x, x.y); }   // (plain loads: the overlap with the neighbouring segment is meant to hit in L2)
    } else {
        const int H = a.H;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const long long g = seg0 + 64 * i + l;
            float2 x = make_float2(0.0f, 0.0f);
            if (g < 0) { if (g + H >= 0) x = a.hist[g + H]; }   // (ROT: the host side hands over the history de-rotated)
            else if (g < a.count) x = a.in[g];
            v[i] = mk2(x.x, x.y);
        }
    }
}

// VOLK's magnitude sawtooth 1 + (g mod 512) gm1 on the INPUT samples (history carries its own); g = seg0 + 64 i + l:
// elements i and i + 8 share their gain
__device__ __forceinline__ void f1k_gain(const FftArgs& a, long long seg0, int l, v2f (&v)[16]) {
    if (a.gm1 == 0.0f) return;
    if (seg0 >= 0) {
        const int base = (int)((seg0 + l) & 511);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float gg = fmaf((float)((base + 64 * i) & 511), a.gm1, 1.0f);
            v[i] = v[i] * mk2(gg, gg);
            v[i + 8] = v[i + 8] * mk2(gg, gg);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const long long g = seg0 + 64 * i + l;
            const float gg = g >= 0 ? fmaf((float)(int)(g & 511), a.gm1, 1.0f) : 1.0f;
            v[i] = v[i] * mk2(gg, gg);
        }
    }
}

// 1024-point transform, spectrum product, inverse: v[i] = element 64 i + l in, filtered element 64 i + l at
// v[rev16(i)] out.  ta(k) / tb(k) / hf(k): the lane's table entries (registers or LDS).  All complex arithmetic is
// packed FP32 (cpk.hip.h): at the one to three waves per SIMD these kernels run at, the 64-bit register pairs cost
// nothing and the halved issue count is kernel time.
template <class TA, class TB, class HF>
__device__ __forceinline__ void f1k_transform(v2f (&v)[16], v2f* ldv, int l, TA ta, TB tb, HF hf) {
    const int kq = l >> 2, j = l & 3;
    v2f* rowq = ldv + kq * kF1P;
    // ---- pass A (over i) + twiddle W1024^(l ka) ------------------------------------------------
    pk_fft16<false>(v);
#pragma unroll
    for (int k = 0; k < 16; k++) ldv[k * kF1P + l] = (k == 0) ? v[rev16(0)] : pk_cmulc<false>(v[rev16(k)], ta(k));
    wave_sync();
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = rowq[j + 4 * m];
    wave_sync();
    // ---- pass B (over m) + twiddle W64^(j kb1) -------------------------------------------------
    pk_fft16<false>(v);
#pragma unroll
    for (int k = 0; k < 16; k++)
        rowq[5 * j + (k >> 2) + 20 * (k & 3)] = (k == 0) ? v[rev16(0)] : pk_cmulc<false>(v[rev16(k)], tb(k));
    // ---- pass C (over j), spectrum product, pass C' --------------------------------------------
    // lane (kq, g = j): entry s*4 + kb0 <-> bin kq + 16 (4 g + s) + 256 kb0
    wave_sync();
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) v[4 * s + jj] = rowq[j + 5 * jj + 20 * s];
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        pk_fft4<false>(v[4 * s], v[4 * s + 1], v[4 * s + 2], v[4 * s + 3]);
#pragma unroll
        for (int k0 = 0; k0 < 4; k0++) v[4 * s + k0] = pk_cmulc<false>(v[4 * s + k0], hf(4 * s + k0));
        pk_fft4<true>(v[4 * s], v[4 * s + 1], v[4 * s + 2], v[4 * s + 3]);
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) rowq[j + 5 * jj + 20 * s] = v[4 * s + jj];
    }
    // ---- pass B' (over kb1) --------------------------------------------------------------------
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const v2f e = rowq[5 * j + (k >> 2) + 20 * (k & 3)];
        v[k] = (k == 0) ? e : pk_cmulc<true>(e, tb(k));
    }
    pk_fft16<true>(v);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 16; m++) rowq[j + 4 * m] = v[rev16(m)];
    wave_sync();
    // ---- pass A' (over ka) ---------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const v2f e = ldv[k * kF1P + l];
        v[k] = (k == 0) ? e : pk_cmulc<true>(e, ta(k));
    }
    pk_fft16<true>(v);
}

// store the valid outputs: element 64 i + l >= ov (at v[rev16(i)]), stream position seg0 + 64 i + l
template <bool ROT>
__device__ __forceinline__ void f1k_store(const FftArgs& a, long long seg0, int l, v2f (&v)[16], v2f q) {
    if (ROT) asm volatile("" : "+v"(q) : "v"(v[0]));   // the 16 output phasors q * wtab[i] are formed HERE, not hoisted above the transform (32 live VGPRs)
    auto rot_out = [&](int i, v2f y) {
        if (ROT) y = pk_cmulc<false>(y, (i == 0) ? q : pk_cmulc<false>(q, mk2(a.wtab[i].x, a.wtab[i].y)));
        return make_float2(y.x, y.y);
    };
    if (ROT || a.strided) {   // (the fused VFO is a resampler: always strided)
        // resampler / VFO: y[n'] sits at stream position n' decm - 1.  One 64-bit division per segment
        // (seg0 + 1 = q0 decm + r0), then a 32-bit multiply-high per element
        const long long s1 = seg0 + 1;
        long long q0 = s1 / a.decm;
        long long r = s1 - q0 * a.decm;
        if (r < 0) { r += a.decm; q0 -= 1; }
        const int r0 = (int)r;
        if (a.m_shift >= 0) {
            // decm | 64: element 64 i + l is an output iff (r0 + l) is a multiple of decm -- the lane keeps all 16 or none
            const unsigned x0 = (unsigned)(r0 + l);
            const long long n0 = q0 + (x0 >> a.m_shift);
            const int per = 64 >> a.m_shift;
            if ((x0 & ((1u << a.m_shift) - 1u)) == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long long n = n0 + i * per;
                    if (64 * i + l >= a.ov && n >= 0 && n < a.nout) a.out[n] = rot_out(i, v[rev16(i)]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int el = 64 * i + l;
                const unsigned x = (unsigned)(r0 + el);
                const unsigned qq = a.decm_inv ? (unsigned)(((unsigned long long)x * a.decm_inv) >> 32) : x / (unsigned)a.decm;
                const long long n = q0 + qq;
                if (el >= a.ov && x - qq * (unsigned)a.decm == 0 && n >= 0 && n < a.nout) a.out[n] = rot_out(i, v[rev16(i)]);
            }
        }
    } else if (seg0 >= 0 && seg0 + kFft1kN <= a.count && seg0 + kFft1kN <= a.nout) {
        float2* __restrict__ o = a.out + seg0 + l;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (64 * i + l >= a.ov) { const float2 y = rot_out(i, v[rev16(i)]); __builtin_nontemporal_store(mk2(y.x, y.y), reinterpret_cast<v2f*>(o + 64 * i)); }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const long long n = seg0 + 64 * i + l;
            if (64 * i + l >= a.ov && n < a.nout) a.out[n] = rot_out(i, v[rev16(i)]);
        }
    }
}

// ---- real data (FIR<float>, PolyphaseResampler<float>): a real filter keeps the real and imaginary parts of its input
// apart, so TWO consecutive real segments ride one complex transform as re / im (as fir_fft_kernel<.., REAL> does with
// its 4096-point segments); loads and stores are 4-byte.  Wave b takes real segments 2b (-> re) and 2b + 1 (-> im).
__device__ __forceinline__ void f1k_hand_over_real(const FftArgs& a, int i) {
    const int H = a.H;
    if (i >= H) return;
    const float* inr = reinterpret_cast<const float*>(a.in);
    const float* hr = reinterpret_cast<const float*>(a.hist);
    float* hn = reinterpret_cast<float*>(a.hist_next);
    const long long g = a.count - H + i;
    hn[i] = g < 0 ? hr[g + H] : inr[g];
}

__device__ __forceinline__ void f1k_load_real(const FftArgs& a, long long segA, long long segB, int l, v2f (&v)[16]) {
    const float* __restrict__ inr = reinterpret_cast<const float*>(a.in);
    const float* __restrict__ hr = reinterpret_cast<const float*>(a.hist);
    if (segA >= 0 && segB + kFft1kN <= a.count) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = mk2(inr[segA + 64 * i + l], inr[segB + 64 * i + l]);
    } else {
        const int H = a.H;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const long long gA = segA + 64 * i + l, gB = segB + 64 * i + l;
            float xa = 0.0f, xb = 0.0f;
            if (gA < 0) { if (gA + H >= 0) xa = hr[gA + H]; }
            else if (gA < a.count) xa = inr[gA];
            if (gB < 0) { if (gB + H >= 0) xb = hr[gB + H]; }
            else if (gB < a.count) xb = inr[gB];
            v[i] = mk2(xa, xb);
        }
    }
}

// one real segment's valid outputs: component C of v (0 = re: segment A, 1 = im: segment B)
template <int C>
__device__ __forceinline__ void f1k_store_real_one(const FftArgs& a, long long seg0, int l, const v2f (&v)[16]) {
    float* __restrict__ outr = reinterpret_cast<float*>(a.out);
    auto val = [&](int i) { return C == 0 ? v[rev16(i)].x : v[rev16(i)].y; };
    if (a.strided) {
        const long long s1 = seg0 + 1;
        long long q0 = s1 / a.decm;
        long long r = s1 - q0 * a.decm;
        if (r < 0) { r += a.decm; q0 -= 1; }
        const int r0 = (int)r;
        if (a.m_shift >= 0) {
            const unsigned x0 = (unsigned)(r0 + l);
            const long long n0 = q0 + (x0 >> a.m_shift);
            const int per = 64 >> a.m_shift;
            if ((x0 & ((1u << a.m_shift) - 1u)) == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long long n = n0 + i * per;
                    if (64 * i + l >= a.ov && n >= 0 && n < a.nout) outr[n] = val(i);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int el = 64 * i + l;
                const unsigned x = (unsigned)(r0 + el);
                const unsigned qq = a.decm_inv ? (unsigned)(((unsigned long long)x * a.decm_inv) >> 32) : x / (unsigned)a.decm;
                const long long n = q0 + qq;
                if (el >= a.ov && x - qq * (unsigned)a.decm == 0 && n >= 0 && n < a.nout) outr[n] = val(i);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const long long n = seg0 + 64 * i + l;
            if (64 * i + l >= a.ov && n < a.nout) outr[n] = val(i);
        }
    }
}

// ---- reference-sized calls: one workgroup = one wave = one segment ----------------------------------------------
template <bool ROT, bool REAL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void fir_fft1k_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[16 * kF1P];
    const int l = threadIdx.x;
    const int nh = (a.H + 63) >> 6;
    if ((int)blockIdx.x < nh) {
        // the hand-over runs in ceil(H / 64) extra workgroups, the FIRST ones of the grid: with the NCO each element costs
        // an FP64 sincos, which would be the tail of the launch if these started last
        if (REAL) f1k_hand_over_real(a, (int)blockIdx.x * 64 + l);
        else f1k_hand_over<ROT>(a, (int)blockIdx.x * 64 + l);
        return;
    }
    const int b = (int)blockIdx.x - nh;
    const long long seg0 = (long long)(REAL ? 2 * b : b) * a.L - a.seg_shift;   // stream position of element 0
    const long long segB = seg0 + a.L;                                           // REAL: the second segment of the pair
    v2f v[16];
    if (REAL) f1k_load_real(a, seg0, segB, l, v);
    else f1k_load(a, seg0, l, v);
    // per-lane constants (L1/L2-resident tables, the same for every wave)
    const int j = l & 3;
    v2f ta[16], tb[16], hf[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const float2 t = a.TA[k * 64 + l]; ta[k] = mk2(t.x, t.y); }
#pragma unroll
    for (int k = 0; k < 16; k++) { const float2 t = a.TB[k * 4 + j]; tb[k] = mk2(t.x, t.y); }
#pragma unroll
    for (int k = 0; k < 16; k++) { const float2 t = a.Hf[k * 64 + l]; hf[k] = mk2(t.x, t.y); }
    v2f q = mk2(1.0f, 0.0f);
    if (ROT) {
        // outputs at position p = seg0 + 64 i + l get exp(j (phase0 + p dphase)) = q * wtab[i]
        const float2 pq = fx_phasor_f32(a.phase0 + (unsigned long long)(seg0 + l) * a.dphase);
        q = mk2(pq.x, pq.y);
        f1k_gain(a, seg0, l, v);
    }
    f1k_transform(v, reinterpret_cast<v2f*>(lds), l, [&](int k) { return ta[k]; }, [&](int k) { return tb[k]; }, [&](int k) { return hf[k]; });
    if (REAL) {
        f1k_store_real_one<0>(a, seg0, l, v);
        f1k_store_real_one<1>(a, segB, l, v);
    } else {
        f1k_store<ROT>(a, seg0, l, v, q);
    }
}

int launch_fir_fft1k(const FftArgs& a, hipStream_t stream) {
    const dim3 grid(a.nblocks + (a.H + 63) / 64);
    if (a.real2) hipLaunchKernelGGL((fir_fft1k_kernel<false, true>), grid, dim3(64), 0, stream, a);
    else if (a.rot) hipLaunchKernelGGL((fir_fft1k_kernel<true>), grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((fir_fft1k_kernel<false>), grid, dim3(64), 0, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
