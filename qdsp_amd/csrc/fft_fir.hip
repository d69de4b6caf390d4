// fft_fir.hip -- the overlap-save FIR kernel (design notes: fft_fir.hip.h).
//
// Tried and rejected (round 1, measured on MI355X, 256 taps, 2^27 samples):
//   * hand-packed complex arithmetic (v_pk_add/mul/fma_f32 with op_sel/neg modifiers through
//     inline asm: 632 packed ops per segment instead of 1329 scalar ones): 0.48-0.51 ms vs
//     0.45 ms.  The opaque asm consumers make hipcc emit read -> s_waitcnt 0 -> use chains
//     (SQ_WAIT_ANY 44 % -> 70 % of wave cycles), which costs more than the halved issue count
//     gains;
//   * the hybrid -- vector-typed add/sub (hipcc emits v_pk_add_f32 itself, no asm) with scalar
//     multiply-by-j and product forms: 1100 VALU ops per segment but 12 spilled VGPRs and 136
//     extra v_mov: 0.54 ms;
//   * two segments per workgroup with every value held as a (segment A, segment B) 2-vector, so
//     hipcc emits v_pk_add/mul/fma_f32 natively (1258 packed ops per TWO segments, no asm):
//     235 VGPRs -> 2 waves/SIMD, 74 KB LDS; numerically identical, 0.48 ms vs 0.45 ms -- the
//     eight barriers per pass are exposed at that occupancy;
//   * the same arithmetic on ext_vector_type(2) values in the forms of cpk.hip.h (hipcc emits
//     v_pk_add/fma_f32 with op_sel and SGPR-pair constants, no shuffles: 702 packed + far fewer
//     scalar ops; this is what the channelizer kernel uses): the 64-bit register pairs cost
//     more VGPRs than the halved issue count buys -- 372 B/lane of scratch at 128 VGPRs
//     (0.86 ms), still 84 B at 168 VGPRs and 3 waves/SIMD (0.55 ms) vs 0.41 ms scalar; the
//     grouped decimators 0.36 / 0.43 ms vs 0.33 / 0.36 ms;
//   * the next segment's loads issued during passes B .. A' of fir_fft_kernel<1> (as the grouped decimator
//     kernel does), at 3 waves/SIMD to make room for the 32 registers: 0.464 vs 0.440 ms on the same box --
//     the fourth wave per SIMD hides more than the prefetch does;
//   * decimate-by-2 through the grouped kernel (pairs of segments, 8-point pruned pass C'): 0.41 ms per 2^26 samples
//     against 0.27 ms for fir_fft_kernel<2>'s per-segment pruned inverse -- with half of the inverse still to do per
//     segment, the grouped kernel's 2 workgroups per CU (70 KB of LDS) cost more than the shared inverse saves;
//   * groups of DEC/2 segments in the grouped kernel (inverse on 128 lanes, 54 KB of LDS, 3 workgroups per CU, no
//     prefetch of the next segment): the 168-VGPR budget spills 170-220 B per lane; decim-8 0.44 ms vs 0.30 ms and the
//     translating decimator 0.46 vs 0.32 ms per 2^27 samples;
//   * 16-byte loads by lane pairs (as fir_fft_kernel does) in the grouped decimator kernel, with
//     segments moved to even starts: 248 VGPRs, decim-8 0.36 ms vs 0.33 ms with 8-byte loads;
//   * non-temporal loads/stores for the sample stream: +-1 %;
//   * pruning the inverse inside one segment for decimators (256/DEC active lanes): no faster
//     than the full inverse -> the grouped kernel below.
//
// Own translation unit because it is compiled with -fno-slp-vectorize: left on, clang's SLP
// pass packs the butterflies' re/im arithmetic into v_pk_add/mul/fma_f32 and pays for it
// with ~450 v_mov/v_xor shuffles per 4096-point block (multiply-by-j swaps) and 209 VGPRs
// (2 waves/SIMD); scalar f32 VALU ops issue at the same FLOP rate on gfx950's 32-wide
// SIMDs once two waves share a SIMD, need no shuffles, and fit in 126 VGPRs (4 waves/SIMD).
#include "fft_fir.hip.h"
#ifndef QDSP_HIP_DIAG
#define QDSP_HIP_DIAG 0
#endif
#include "cfft.hip.h"
#include "cpk.hip.h"
#include "ldsdma.hip.h"

namespace qk {

typedef float f4v __attribute__((ext_vector_type(4)));

// Column of element e in LDS layout 1 ([k0][element], row pitch kFftRow1 = 272): even elements
// first, odd elements from column 136.  Keeps the pass-A writes (lanes own elements 2l', then
// 2l'+1) contiguous and the pass-B reads (16 consecutive elements per 16 lanes) conflict-free:
// evens on banks 16j.., odds +16, the next k0 row (+544 dwords) +32.
__host__ __device__ constexpr int pos1(int e) { return (e >> 1) + 136 * (e & 1); }

// Overlap-save block filter (see fft_fir.hip.h):
//   DEC == 1 : FIR<complex_t>            y[n]  = sum_k taps[k] s[n - (N-1) + k]
//   DEC  > 1 : PolyphaseResampler, interp 1, decim DEC (DEC | 16)
//                                         y[n'] = sum_k taps[k] s[n'*DEC - P + k]
//   ROT      : FrequencyXlator applied to `in` while loading (the fused VFO)
// s = hist ++ in.  Segment b starts at stream position b*L - seg_shift; its elements
// i >= ov are valid filter outputs (position p = seg0 + i).  For DEC > 1 the segments are
// placed so that the wanted positions (p == -1 mod DEC) are the elements with i == 0 mod DEC,
// i.e. the last radix-16 digit n0 is a multiple of DEC: the inverse transform keeps only
// those 16/DEC values of n0, and passes B'/A' run on 256/DEC lanes.
//   REAL     : FIR<float> / PolyphaseResampler<float> (real samples, real taps): a real filter keeps the
//              real and imaginary parts of its input apart, so TWO consecutive real segments ride one
//              complex transform as re / im (segment 2b -> re, 2b+1 -> im); loads and stores are 4-byte.
template <int DEC, bool ROT, bool REAL = false>
__global__ __launch_bounds__(kFftNT, (DEC == 1 || !ROT) ? 4 : 3) void fir_fft_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[kFftLdsElems + 16 * 17];
    // ROT: exp(j 2pi (phase0 + seg0 dphase)) of the workgroup's current segment lives HERE, not in four VGPRs of every lane (round 3: the fused
    // variants ran 147 VGPRs = three workgroups per CU where the plain FIR has 128 = four; with this and the output phasor formed at the
    // stores instead of at the loads they fit 128 as well).  Two slots: segment k reads [k & 1], lane 0 leaves segment k+1's in the other.
    __shared__ double2 s_pb[2];
    float2* tbl = lds + kFftLdsElems;  // pass-B twiddles W256^(lo*k), rows padded to 17
    const int t_lane = threadIdx.x, t = t_lane;
    const int H = a.H;
    // Element of the segment this lane owns in passes A / A' (all 16 rows n2 of it).  Lanes l and
    // l+32 of a wave own the adjacent elements 2l', 2l'+1 so that one 16-byte load/store covers
    // both: each lane moves float4s for half the rows and trades halves with v_permlane32_swap.
    const int te_lane = (t & ~63) | ((t & 31) << 1) | ((t >> 5) & 1);
    const int te = te_lane;
    constexpr int NS = 16 / DEC;       // kept values of n0: 0, DEC, 2*DEC, ...
    constexpr int NACT = 256 / DEC;    // lanes active in the pruned inverse passes

    if ((int)blockIdx.x == a.nwg) {
        // history hand-over (filter.h:71 / resampling.h:129): last H samples of hist ++ in
        if constexpr (REAL) {
            const float* inr = reinterpret_cast<const float*>(a.in);
            const float* hr = reinterpret_cast<const float*>(a.hist);
            float* hn = reinterpret_cast<float*>(a.hist_next);
            for (int i = t; i < H; i += kFftNT) {
                const long long g = a.count - H + i;
                hn[i] = g < 0 ? hr[g + H] : inr[g];
            }
            return;
        }
        for (int i = t; i < H; i += kFftNT) {
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
                    // the un-rotated form (gain kept) for the next overlap-save call: saves its de-rotation launch
                    if (a.hist_raw_next) a.hist_raw_next[i] = make_float2(v.x * gain, v.y * gain);
                    v = cmulc<false>(v, make_float2((float)p.x * gain, (float)p.y * gain));
                }
            }
            a.hist_next[i] = v;
        }
        return;
    }

    // per-lane constants, loaded once
    // DEC == 1 keeps both tables in VGPRs (124 in all).  The pruned-inverse variants need a
    // few more live values, so they re-read the pass-A twiddles from the L2-resident table
    // every block (opaque pointer below) rather than spill.
    float2 ta[16], hf[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hf[k] = a.Hf[t * 16 + k];
        if (DEC == 1) ta[k] = a.TA[te * 16 + k];
    }
    // pass-A' twiddles of the pruned inverse: lane w = n1*NS + s <-> element n1*16 + s*DEC.
    // Re-read from the (L2-resident) table every block by the 256/DEC active lanes instead of
    // living in 32 more VGPRs: keeps the kernel at 128 VGPRs = 4 waves/SIMD without spills.
    const int e0 = ((t / NS) * 16 + (t % NS) * DEC) & 255;
    tbl[(t >> 4) * 17 + (t & 15)] = a.TB[t];

    float2 pl = make_float2(1.0f, 0.0f);   // exp(j 2pi e dphase), e = element of the lane's outputs
    if (ROT) {
        const double2 p = fx_phasor((unsigned long long)(DEC == 1 ? te : e0) * a.dphase);
        pl = make_float2((float)p.x, (float)p.y);
        if (t == 0) s_pb[0] = fx_phasor(a.phase0 + (unsigned long long)((long long)blockIdx.x * a.L - a.seg_shift) * a.dphase);
    }
    int seg_parity = 0;

    for (int b = blockIdx.x; b < a.nblocks; b += a.nwg) {
        // (ROT: the lane's element index is made opaque once per segment, so that the 64-bit per-lane addresses built from it are formed here
        // -- one v_lshl_add_u64 each -- instead of being hoisted out of the loop, held in eight VGPRs and spilled: the fourth workgroup per CU)
        int t_seg = t_lane;
        if constexpr (DEC == 1 ? (ROT || REAL) : !ROT) asm volatile("" : "+v"(t_seg));
        const int t = t_seg, hi = t >> 4, lo = t & 15;
        const int te = (t & ~63) | ((t & 31) << 1) | ((t >> 5) & 1), half = (t >> 5) & 1, pte = pos1(te);
        const float2* tb = tbl + lo * 17;
        const long long seg0 = (long long)(REAL ? 2 * b : b) * a.L - a.seg_shift;  // stream position of element 0
        const long long segB = seg0 + a.L;                         // REAL: the second segment of the pair
        const bool interior = seg0 >= 0 && (REAL ? segB : seg0) + kFftN <= a.count;
        float2 v[16];
        // ---- load: the lane ends up with elements n2*256 + te, n2 = 0..15 ------------------
        if constexpr (REAL) {
            const float* __restrict__ inr = reinterpret_cast<const float*>(a.in);
            const float* __restrict__ hr = reinterpret_cast<const float*>(a.hist);
            if (interior && a.vec) {
                // 8-byte loads, as the complex path does with 16-byte ones: lanes l and l+32 own the adjacent elements
                // (te & ~1, +1); each loads both for half the rows (of segment A and of segment B) and trades
                // halves with v_permlane32_swap
                const float2* __restrict__ pa = reinterpret_cast<const float2*>(inr + seg0 + (te & ~1)) + half * 8 * 128;
                const float2* __restrict__ pb = reinterpret_cast<const float2*>(inr + segB + (te & ~1)) + half * 8 * 128;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float2 qa = pa[r * 128], qb = pb[r * 128];
                    const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa.x), __float_as_uint(qa.y), false, false);
                    const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(qb.x), __float_as_uint(qb.y), false, false);
                    v[r] = make_float2(__uint_as_float(sa[0]), __uint_as_float(sb[0]));
                    v[8 + r] = make_float2(__uint_as_float(sa[1]), __uint_as_float(sb[1]));
                }
            } else if (interior) {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) v[n2] = make_float2(inr[seg0 + n2 * 256 + te], inr[segB + n2 * 256 + te]);
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const long long gA = seg0 + n2 * 256 + te, gB = segB + n2 * 256 + te;
                    float xa = 0.0f, xb = 0.0f;
                    if (gA < 0) { if (gA + H >= 0) xa = hr[gA + H]; }
                    else if (gA < a.count) xa = inr[gA];
                    if (gB < 0) { if (gB + H >= 0) xb = hr[gB + H]; }
                    else if (gB < a.count) xb = inr[gB];
                    v[n2] = make_float2(xa, xb);
                }
            }
        } else if (interior && a.vec) {
            // rows half*8 + r, elements (te & ~1, +1); row pitch 256 samples = 128 float4
            const float4* __restrict__ p4 = reinterpret_cast<const float4*>(a.in + seg0 + (te & ~1)) + half * 8 * 128;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float4 q4 = p4[r * 128];
                // x/y regs: lanes 32-63 <-> z/w regs of lanes 0-31: afterwards (x,y) = row r and
                // (z,w) = row 8+r of the lane's OWN element, for every lane
                const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.x), __float_as_uint(q4.z), false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.y), __float_as_uint(q4.w), false, false);
                v[r] = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
                v[8 + r] = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
            }
        } else if (interior) {
            const float2* __restrict__ p = a.in + seg0 + te;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) v[n2] = p[n2 * 256];
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const long long g = seg0 + n2 * 256 + te;
                float2 x = make_float2(0.0f, 0.0f);
                if (g < 0) { if (g + H >= 0) x = a.hist[g + H]; }   // (ROT: the host side hands over the history de-rotated)
                else if (g < a.count) x = a.in[g];
                v[n2] = x;
            }
        }
        // ROT: VOLK's magnitude sawtooth 1 + (g mod 512)*gm1 stays on the INPUT samples (a real scale; it
        // takes two values per lane and segment at g = seg0 + te + 256 n2, since adding 256 toggles bit 8
        // of g mod 512); the phasor goes on the kept OUTPUTS (positions seg0 + e + 256 n2).
        float2 q = make_float2(1.0f, 0.0f);
        if (ROT) {
            if (a.gm1 != 0.0f) {
                const int gb = (int)((seg0 + te) & 511);
                const float g0 = fmaf((float)gb, a.gm1, 1.0f), g1 = fmaf((float)(gb ^ 256), a.gm1, 1.0f);
                const bool nohist = seg0 >= 0;
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const float gg = (n2 & 1) ? g1 : g0;
                    if (nohist || seg0 + n2 * 256 + te >= 0) v[n2] = make_float2(v[n2].x * gg, v[n2].y * gg);   // (history carries its gain)
                }
            }
        }
        // the output phasor of this lane for this segment, formed where the stores begin (the segment's barriers lie between lane 0's
        // write of a slot and every read of it)
        auto make_q = [&]() {
            if (!ROT) return;
            const double2 pbv = s_pb[seg_parity];
            q = cmulc<false>(make_float2((float)pbv.x, (float)pbv.y), pl);
            if (t == 0) s_pb[seg_parity ^ 1] = dcmul(pbv, a.rot_step);
            seg_parity ^= 1;
        };
        auto rot_out = [&](int n2, float2 y) {
            if (!ROT) return y;
            return cmulc<false>(y, (n2 == 0) ? q : cmulc<false>(q, a.wtab[n2]));
        };
        // ---- pass A (over n2) + twiddle W4096^(t*k0) -----------------------------------
        fft16<false>(v);
        if constexpr (DEC > 1) {
            const float2* tap = a.TA + te * 16;
            asm volatile("" : "+v"(tap) : "v"(v[0].x));  // opaque + ordered after the butterflies
#pragma unroll
            for (int k = 1; k < 16; k++) ta[k] = tap[k];
        }
        __syncthreads();  // previous block's last LDS reads are done
#pragma unroll
        for (int k = 0; k < 16; k++) lds[k * kFftRow1 + pte] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], ta[k]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[hi * kFftRow1 + pos1(j * 16 + lo)];
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
        float2 y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cmulc<false>(v[rev16(k)], hf[k]);
        fft16<true>(y);
        __syncthreads();
        if constexpr (DEC == 1) {
#pragma unroll
            for (int j = 0; j < 16; j++)
                lds[t * kFftRow2 + j] = (j == 0) ? y[rev16(0)] : cmulc<true>(y[rev16(j)], tb[j]);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = lds[(hi * 16 + j) * kFftRow2 + lo];
            // ---- pass B' (over k1) ---------------------------------------------------------
            fft16<true>(v);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) lds[hi * kFftRow1 + pos1(j * 16 + lo)] = v[rev16(j)];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float2 e = lds[k * kFftRow1 + pte];
                v[k] = (k == 0) ? e : cmulc<true>(e, ta[k]);
            }
            // ---- pass A' (over k0) and store the L valid outputs --------------------------
            fft16<true>(v);
            make_q();
            const long long o0 = seg0 + te;  // output index == stream position (element n2*256 + te)
            // strided stores: element i of a segment starting at s sits at position p = s + i and is output
            // n' iff p + 1 == n' * decm.  One 64-bit division per segment (s + 1 = q0 * decm + r0), then per
            // element x = r0 + i < decm + 4096 goes through a 32-bit multiply-high (16 64-bit divisions per
            // lane and segment were ~20 % of the kernel's instructions).
            auto seg_split = [&](long long s1, long long& q0, int& r0) {
                q0 = s1 / a.decm;
                long long r = s1 - q0 * a.decm;
                if (r < 0) { r += a.decm; q0 -= 1; }
                r0 = (int)r;
            };
            auto out_index = [&](long long q0, int r0, int i, long long& n) -> bool {   // true: element i is an output, n its index
                const unsigned x = (unsigned)(r0 + i);
                const unsigned q = a.decm_inv ? (unsigned)(((unsigned long long)x * a.decm_inv) >> 32) : x / (unsigned)a.decm;
                n = q0 + q;
                // (n >= 0: at decm == 1 the first segment's element at stream position -2 would be "output -1")
                return x - q * (unsigned)a.decm == 0 && n >= 0;
            };
            if constexpr (REAL) {
                float* __restrict__ outr = reinterpret_cast<float*>(a.out);
                long long qA = 0, qB = 0;
                int rA = 0, rB = 0;
                if (a.strided) {
                    seg_split(seg0 + 1, qA, rA);
                    seg_split(segB + 1, qB, rB);
                }
                if (!a.strided && interior && a.vec && segB + kFftN <= a.nout) {
                    // FIR<float>: the mirror image of the loads -- 8-byte stores of both segments
                    float2* __restrict__ oa = reinterpret_cast<float2*>(outr + seg0 + (te & ~1)) + half * 8 * 128;
                    float2* __restrict__ ob = reinterpret_cast<float2*>(outr + segB + (te & ~1)) + half * 8 * 128;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const float2 lo_row = v[rev16(r)], hi_row = v[rev16(8 + r)];
                        const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_row.x), __float_as_uint(hi_row.x), false, false);
                        const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_row.y), __float_as_uint(hi_row.y), false, false);
                        if ((half * 8 + r) * 256 + (te & ~1) >= a.ov) {   // ov is even: both elements or neither
                            oa[r * 128] = make_float2(__uint_as_float(sa[0]), __uint_as_float(sa[1]));
                            ob[r * 128] = make_float2(__uint_as_float(sb[0]), __uint_as_float(sb[1]));
                        }
                    }
                } else
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    if (n2 * 256 + te < a.ov) continue;
                    const float2 y = v[rev16(n2)];
                    const long long pA = o0 + n2 * 256, pB = pA + a.L;
                    if (a.strided) {    // resampler: y[n'] sits at stream position n'*decm - 1
                        long long nA, nB;
                        if (out_index(qA, rA, n2 * 256 + te, nA) && nA < a.nout) outr[nA] = y.x;
                        if (out_index(qB, rB, n2 * 256 + te, nB) && nB < a.nout) outr[nB] = y.y;
                    } else {
                        if (pA < a.nout) outr[pA] = y.x;
                        if (pB < a.nout) outr[pB] = y.y;
                    }
                }
            } else if (a.strided) {
                // Any integer decimation: the full inverse ran; keep the positions p == -1 (mod decm)
                // (y[n'] sits at stream position n'*decm - 1) -- strided 8-byte stores, 1/decm of them.
                long long q0;
                int r0;
                seg_split(seg0 + 1, q0, r0);
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    long long n;
                    if (n2 * 256 + te >= a.ov && out_index(q0, r0, n2 * 256 + te, n) && n < a.nout) a.out[n] = rot_out(n2, v[rev16(n2)]);
                }
            } else if (interior && a.vec) {
                float4* __restrict__ o4 = reinterpret_cast<float4*>(a.out + seg0 + (te & ~1)) + half * 8 * 128;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float2 lo_row = v[rev16(r)], hi_row = v[rev16(8 + r)];
                    const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_row.x), __float_as_uint(hi_row.x), false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_row.y), __float_as_uint(hi_row.y), false, false);
                    // now (x,y | z,w) = elements (te & ~1, +1) of row half*8 + r
                    if ((half * 8 + r) * 256 + (te & ~1) >= a.ov)   // ov is even: both elements or neither
                        o4[r * 128] = make_float4(__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sx[1]), __uint_as_float(sy[1]));
                }
            } else if (interior) {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++)
                    if (n2 * 256 + te >= a.ov) a.out[o0 + n2 * 256] = v[rev16(n2)];
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const long long n = o0 + n2 * 256;
                    if (n2 * 256 + te >= a.ov && n < a.nout) a.out[n] = v[rev16(n2)];
                }
            }
        } else {
            // ---- pruned inverse: keep n0 = s*DEC only ---------------------------------------
            // lane (k0 = hi, k1 = lo) -> row (k0*NS + s), column k1
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int n0 = s * DEC;
                lds[(hi * NS + s) * kFftRow2 + lo] = (n0 == 0) ? y[rev16(0)] : cmulc<true>(y[rev16(n0)], tb[n0]);
            }
            __syncthreads();
            if (t < NACT) {  // lane u = k0*NS + s : pass B' over k1
#pragma unroll
                for (int j = 0; j < 16; j++) v[j] = lds[t * kFftRow2 + j];
                fft16<true>(v);
            }
            __syncthreads();
            if (t < NACT) {  // -> row (n1*NS + s), column k0
                const int k0 = t / NS, s = t % NS;
#pragma unroll
                for (int j = 0; j < 16; j++) lds[(j * NS + s) * kFftRow2 + k0] = v[rev16(j)];
            }
            __syncthreads();
            if (t < NACT) {  // lane w = n1*NS + s : twiddle, pass A' over k0, store
                const float2* ta2 = a.TA + e0 * 16;
                asm volatile("" : "+v"(ta2) : "v"(v[0].x));  // opaque: not hoisted out of the block loop, not issued early
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float2 e = lds[t * kFftRow2 + k];
                    v[k] = (k == 0) ? e : cmulc<true>(e, ta2[k]);
                }
                fft16<true>(v);
                make_q();
                // (32-bit lane offsets from a wave-uniform base: sixteen 64-bit per-lane indices cost the ROT variants two spilled VGPRs)
                const long long nb = ((long long)b * a.L - a.ov) / DEC;  // (b*L + i - ov)/DEC at i = 0
                float2* __restrict__ seg_out = a.out + nb;
                const long long left = a.nout - nb;
                const int end_out = left < kFftN ? (int)left : kFftN;      // (i / DEC < 4096 / DEC)
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const int i = n2 * 256 + e0;
                    if (i >= a.ov && i / DEC < end_out) seg_out[i / DEC] = rot_out(n2, v[rev16(n2)]);
                }
            }
        }
    }
}

// ---- fir_fft_dma_kernel (round 3): FIR<complex_t>, interior segments fed by LDS-DMA ------------------------------
// The same transform as fir_fft_kernel<1> (passes, tables, rounding: bit-identical results), re-arranged around what the
// ablation of that kernel showed (profiles/r03_ablate_fir_fft.txt): arithmetic + LDS + barriers alone take 0.335 ms per
// 2^27 samples, the loads add 0.05 ms and the stores 0.08 ms -- a workgroup's wait for its next segment (vmcnt counts in
// order) also waits until the eight stores it has just issued are acknowledged.  Here
//   * the NEXT segment is requested before this segment's last pass: `global_load_lds_dwordx4` straight into the exchange
//     buffer (no VGPRs: the kernel stays at 4 waves per SIMD), issued right after the last read of the inverse (r4), so it
//     travels under pass A' and the stores, and is OLDER than those stores in the vmcnt order: the wait at the top of the
//     next segment is `vmcnt(8)` -- the eight stores stay in flight and are never waited for;
//   * a lane owns one column of the segment (elements te + 256 n2) in passes A / A'; a wave only ever touches ITS 64 columns
//     between the last barrier of a segment and the first of the next, so the DMA (sixteen 512-byte half-wave requests,
//     one per row, into the wave's own columns), the raw read-back and the in-place write of pass A need no barrier:
//     seven barriers per segment instead of eight, and none of them drains the vector-memory counter (raw s_barrier behind
//     an LDS-only wait; `__syncthreads()` would wait for the DMA and the stores);
//   * layout 1 is plain [row][column] (the DMA delivers pairs of samples, so fir_fft_kernel's even/odd split is not
//     available): lanes l / l + 16 own adjacent columns, which keeps every ds_read_b64 group on one contiguous 256-byte run
//     (the in-place ds_write_b64 of pass A is 2-way conflicted: 8 LDS cycles against the 6 the instruction takes anyway)
//     and lets a v_permlane16_swap build the 16-byte stores.
// Segments that touch the history or the end of the input take guarded loads into registers as in fir_fft_kernel.
// The DMA instruction is inline asm (it must not enter hipcc's own vmcnt bookkeeping, which would drain it together with
// the stores at the next barrier): M0 is saved and restored inside the statement (cdna_hip_programming.md 5.7).
#define QK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(kFftNT, 4) void fir_fft_dma_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[kFftLdsElems + 16 * 17];
    float2* tbl = lds + kFftLdsElems;  // pass-B twiddles W256^(lo*k), rows padded to 17
    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int H = a.H;
    // Column of the segment this lane owns in passes A / A' (elements te + 256 n2).  Lanes l and l + 16 own the adjacent
    // columns 2c, 2c + 1, so that lanes 0-15 / 16-31 of a half-wave read the even / odd elements of sixteen consecutive
    // pairs (one contiguous 256-byte run per ds_read_b64 group) and a v_permlane16_swap pairs them up for 16-byte stores.
    const int half = (t >> 4) & 1;
    const int te = (t & ~31) | ((t & 15) << 1) | half;

    if ((int)blockIdx.x == a.nwg) {   // history hand-over (filter.h:71): last H samples of hist ++ in
        for (int i = t; i < H; i += kFftNT) {
            const long long g = a.count - H + i;
            a.hist_next[i] = g < 0 ? a.hist_keep[g + H] : a.in[g];
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
    // the table loads are waited for HERE: left to hipcc, the wait (vmcnt(0): stores and DMA included) lands at their
    // first use, inside the segment loop
    {
        float touch = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k++) touch += hf[k].x + hf[k].y + (k ? ta[k].x + ta[k].y : 0.0f);
        asm volatile("" ::"v"(touch));
    }
    __syncthreads();

    // this wave's 64 columns of row 0 as an LDS byte address, its first column, and the lane's byte offset in a row request
    const int wcol = __builtin_amdgcn_readfirstlane(t & ~63);
    const unsigned lds_cols = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds) + (unsigned)wcol * 8u;
    const unsigned voff = (unsigned)(t & 31) * 16u;
    auto seg_start = [&](int b) { return (long long)b * a.L - a.seg_shift; };
    auto is_interior = [&](int b) {
        const long long s0 = seg_start(b);
        return b < a.nblocks && s0 >= 0 && s0 + kFftN <= a.count;
    };
    auto request = [&](int b) {   // rows 0..15 of segment b -> this wave's columns (lanes 0-31: 512 bytes per row)
        const float2* src = a.in + seg_start(b) + wcol;
        if ((t & 32) == 0) {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) dma16_to_lds(src + n2 * 256, voff, lds_cols + (unsigned)(n2 * kFftRow1) * 8u);
        }
    };

    int b = blockIdx.x;
    bool landed_in_lds = false, eight_younger = false;
    if (b < a.nblocks && is_interior(b)) {
        request(b);
        landed_in_lds = true;
    }
    for (; b < a.nblocks; b += a.nwg) {
        const long long seg0 = seg_start(b);
        float2 v[16];
        if (landed_in_lds) {
            if (eight_younger) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // first / last segments: guarded loads, parked in the wave's own columns so that both kinds of segment continue
            // through the same read-back (hipcc's wait-count pass merges the two paths: loads still pending at the join
            // would put an `s_waitcnt vmcnt(0)` -- stores included -- in front of the fast path's read-back)
            // (all per-lane indices are 32-bit offsets from wave-uniform bases: 64-bit per-lane pointers hoisted out of
            // the segment loop were what spilled)
            const int first_in = seg0 < 0 ? (int)-seg0 : 0;                                          // elements below: history / zeros
            const int end_in = a.count - seg0 < kFftN ? (int)(a.count - seg0) : kFftN;               // elements from here on: zeros
            const float2* __restrict__ seg_in = a.in + seg0;
            const float2* __restrict__ seg_hist = a.hist + (H - first_in);                           // element i < first_in -> hist[i - first_in + H]
            int tev = te;
            asm volatile("" : "+v"(tev));   // opaque: keeps hipcc from hoisting sixteen 64-bit per-lane addresses out of the segment loop
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const int i = n2 * 256 + tev;
                float2 x = make_float2(0.0f, 0.0f);
                if (i < first_in) { if (i - first_in + H >= 0) x = seg_hist[i]; }
                else if (i < end_in) x = seg_in[i];
                lds[n2 * kFftRow1 + te] = x;
            }
        }
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = lds[n2 * kFftRow1 + te];
        // ---- pass A (over n2) + twiddle W4096^(t*k0), written back in place (the wave's own columns) ------------
        fft16<false>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[k * kFftRow1 + te] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], ta[k]);
        QK_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[hi * kFftRow1 + j * 16 + lo];
        // ---- pass B (over n1) + twiddle W256^(n0*k1) -----------------------------------------------------------
        fft16<false>(v);
        QK_LDS_BARRIER();
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[(hi * 16 + k) * kFftRow2 + lo] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], tb[k]);
        QK_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[t * kFftRow2 + j];
        // ---- pass C (over n0), spectrum * Hf, pass C' (over k2) ---------------------------------------------------
        fft16<false>(v);
        float2 y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cmulc<false>(v[rev16(k)], hf[k]);
        fft16<true>(y);
        QK_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; j++)
            lds[t * kFftRow2 + j] = (j == 0) ? y[rev16(0)] : cmulc<true>(y[rev16(j)], tb[j]);
        QK_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[(hi * 16 + j) * kFftRow2 + lo];
        // ---- pass B' (over k1) --------------------------------------------------------------------------------------
        fft16<true>(v);
        QK_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; j++) lds[hi * kFftRow1 + j * 16 + lo] = v[rev16(j)];
        QK_LDS_BARRIER();
        float2 e[16];
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = lds[k * kFftRow1 + te];
        // ---- the next segment's samples: requested now, into the columns this wave has just read ---------------------
        const int nb = b + a.nwg;
        landed_in_lds = is_interior(nb);
        if (landed_in_lds) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads above have returned
            request(nb);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = (k == 0) ? e[k] : cmulc<true>(e[k], ta[k]);
        // ---- pass A' (over k0) and store the L valid outputs --------------------------------------------------------
        fft16<true>(v);
        eight_younger = false;
        if (seg0 >= 0 && seg0 + kFftN <= a.nout) {
            // 16-byte stores: the lane with half == 0 stores both columns of its pair for rows 0-7, its partner for rows 8-15
            // (one wave instruction = four 256-byte runs: two of row r, two of row r + 8)
            f4v* __restrict__ o4 = reinterpret_cast<f4v*>(a.out + seg0);   // wave-uniform; 16-byte units
            int pair = (te >> 1) + half * 8 * 128;
            asm volatile("" : "+v"(pair));   // (as above)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float2 lo_row = v[rev16(r)], hi_row = v[rev16(8 + r)];
                const auto sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo_row.x), __float_as_uint(hi_row.x), false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo_row.y), __float_as_uint(hi_row.y), false, false);
                // now (x,y | z,w) = columns (te & ~1, +1) of row half*8 + r
                if ((half * 8 + r) * 256 + (te & ~1) >= a.ov)   // ov is even: both elements or neither
                    o4[pair + r * 128] = (f4v){__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sx[1]), __uint_as_float(sy[1])};
            }
            eight_younger = true;   // (rows 8..15 always lie past the overlap: ov <= 2048 on this path, so all eight are issued)
        } else {
            const int first_out = seg0 < 0 ? (int)-seg0 : 0;                                          // output index >= 0
            const int from = a.ov > first_out ? a.ov : first_out;
            const int end_out = a.nout - seg0 < kFftN ? (int)(a.nout - seg0) : kFftN;
            float2* __restrict__ seg_out = a.out + seg0;
            int tev = te;
            asm volatile("" : "+v"(tev));   // (as above)
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const int i = n2 * 256 + tev;
                if (i >= from && i < end_out) seg_out[i] = v[rev16(n2)];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- fir_fft_dmapk_kernel: the same kernel with its complex arithmetic on packed FP32 pairs -------------------------------
// PMC of fir_fft_dma_kernel (profiles/r03_pmc_fir_fft_dma_kernel.json): waits 39 % -> 20 % of the wave cycles, issue stalls 30 % ->
// 45 %: with the memory waits gone the kernel is VALU-issue-bound at ~4.1 cycles per instruction (5261 instructions per segment).
// Packed instructions do two lanes' worth per issue: v_pk_add_f32 for the butterflies' adds, and a complex product is two
// instructions -- v_pk_mul_f32 + v_pk_fma_f32 with the swap on op_sel and the ONE-lane sign on neg_lo / neg_hi (inline asm:
// hipcc folds whole-vector negation only) -- instead of 2 v_mul + 2 v_fmac.
// STAMPS: diagnostic build (never shipped to a caller: launch_fir_fft takes it only when FftArgs::stamps is set, which only
// scripts/stamp_fir_fft.py does): every wave sums, per phase of the segment loop, the shader-clock ticks (s_memtime) it
// spent there; wave 0 of each workgroup writes its sixteen sums to stamps[blockIdx.x * 16 ..].  Nothing is computed from them.
template <bool CONJ, int ABL> __device__ __forceinline__ v2f abl_cmul(v2f a, v2f b) {
    if (ABL & 1) return a;
    return pk_cmul2<CONJ>(a, b);
}

// ABL: ablations for profiles/r03_ablate_fir_fft_dmapk.txt (diagnostic builds): 1 = no arithmetic (the LDS exchanges, barriers, DMA and stores
// stay), 2 = no global stores, 4 = no DMA (the segments are whatever lies in LDS).
template <bool STAMPS, int ABL = 0>
__global__ __launch_bounds__(kFftNT, 4) void fir_fft_dmapk_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) v2f lds[kFftLdsElems + 16 * 17];
    v2f* tbl = lds + kFftLdsElems;  // pass-B twiddles W256^(lo*k), rows padded to 17
    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int H = a.H;
    // Column of the segment this lane owns in passes A / A' (elements te + 256 n2).  Lanes l and l + 16 own the adjacent
    // columns 2c, 2c + 1, so that lanes 0-15 / 16-31 of a half-wave read the even / odd elements of sixteen consecutive
    // pairs (one contiguous 256-byte run per ds_read_b64 group) and a v_permlane16_swap pairs them up for 16-byte stores.
    const int half = (t >> 4) & 1;
    const int te = (t & ~31) | ((t & 15) << 1) | half;

    if ((int)blockIdx.x == a.nwg) {   // history hand-over (filter.h:71): last H samples of hist ++ in
        for (int i = t; i < H; i += kFftNT) {
            const long long g = a.count - H + i;
            a.hist_next[i] = g < 0 ? a.hist_keep[g + H] : a.in[g];
        }
        return;
    }
    v2f ta[16], hf[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hf[k] = reinterpret_cast<const v2f*>(a.Hf)[t * 16 + k];
        ta[k] = reinterpret_cast<const v2f*>(a.TA)[te * 16 + k];
    }
    tbl[(t >> 4) * 17 + (t & 15)] = reinterpret_cast<const v2f*>(a.TB)[t];
    const v2f* tb = tbl + lo * 17;
    // the table loads are waited for HERE: left to hipcc, the wait (vmcnt(0): stores and DMA included) lands at their
    // first use, inside the segment loop
    {
        float touch = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k++) touch += hf[k].x + hf[k].y + (k ? ta[k].x + ta[k].y : 0.0f);
        asm volatile("" ::"v"(touch));
    }
    __syncthreads();

    // this wave's 64 columns of row 0 as an LDS byte address, its first column, and the lane's byte offset in a row request
    const int wcol = __builtin_amdgcn_readfirstlane(t & ~63);
    const unsigned lds_cols = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds) + (unsigned)wcol * 8u;
    const unsigned voff = (unsigned)(t & 31) * 16u;
    auto seg_start = [&](int b) { return (long long)b * a.L - a.seg_shift; };
    auto is_interior = [&](int b) {
        const long long s0 = seg_start(b);
        return b < a.nblocks && s0 >= 0 && s0 + kFftN <= a.count;
    };
    auto request = [&](int b) {   // rows 0..15 of segment b -> this wave's columns (lanes 0-31: 512 bytes per row)
        const float2* src = a.in + seg_start(b) + wcol;
        if ((t & 32) == 0 && !(ABL & 4)) {
            if (a.nt & 1) {   // rows 0 and 15 hold the overlap the neighbouring segments read as well: those stay cacheable
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    if (n2 == 0 || n2 == 15) dma16_to_lds(src + n2 * 256, voff, lds_cols + (unsigned)(n2 * kFftRow1) * 8u);
                    else dma16_to_lds_nt(src + n2 * 256, voff, lds_cols + (unsigned)(n2 * kFftRow1) * 8u);
                }
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) dma16_to_lds(src + n2 * 256, voff, lds_cols + (unsigned)(n2 * kFftRow1) * 8u);
            }
        }
    };

    unsigned acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick = 0;
    if constexpr (STAMPS) tick = __builtin_amdgcn_s_memtime();
#define QK_STAMP(i)                                                          \
    if constexpr (STAMPS) {                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        acc[i] += (unsigned)(now_ - tick);                                   \
        tick = now_;                                                         \
    }
    int b = blockIdx.x;
    bool landed_in_lds = false, eight_younger = false;
    if (b < a.nblocks && is_interior(b)) {
        request(b);
        landed_in_lds = true;
    }
    for (; b < a.nblocks; b += a.nwg) {
        const long long seg0 = seg_start(b);
        v2f v[16];
        QK_STAMP(15)   // pass A', stores, loop control
        if (landed_in_lds) {
            if (eight_younger) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QK_STAMP(0)   // wait for the DMA
        } else {
            // first / last segments: guarded loads, parked in the wave's own columns so that both kinds of segment continue
            // through the same read-back (hipcc's wait-count pass merges the two paths: loads still pending at the join
            // would put an `s_waitcnt vmcnt(0)` -- stores included -- in front of the fast path's read-back)
            // (all per-lane indices are 32-bit offsets from wave-uniform bases: 64-bit per-lane pointers hoisted out of
            // the segment loop were what spilled)
            const int first_in = seg0 < 0 ? (int)-seg0 : 0;                                          // elements below: history / zeros
            const int end_in = a.count - seg0 < kFftN ? (int)(a.count - seg0) : kFftN;               // elements from here on: zeros
            const float2* __restrict__ seg_in = a.in + seg0;
            const float2* __restrict__ seg_hist = a.hist + (H - first_in);                           // element i < first_in -> hist[i - first_in + H]
            int tev = te;
            asm volatile("" : "+v"(tev));   // opaque: keeps hipcc from hoisting sixteen 64-bit per-lane addresses out of the segment loop
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const int i = n2 * 256 + tev;
                float2 x = make_float2(0.0f, 0.0f);
                if (i < first_in) { if (i - first_in + H >= 0) x = seg_hist[i]; }
                else if (i < end_in) x = seg_in[i];
                lds[n2 * kFftRow1 + te] = mk2(x.x, x.y);
            }
        }
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = lds[n2 * kFftRow1 + te];
        // ---- pass A (over n2) + twiddle W4096^(t*k0), written back in place (the wave's own columns) ------------
        if (!(ABL & 1)) pk_fft16<false>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[k * kFftRow1 + te] = (k == 0) ? v[rev16(0)] : abl_cmul<false, ABL>(v[rev16(k)], ta[k]);
        QK_STAMP(1)
        QK_LDS_BARRIER();
        QK_STAMP(2)
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[hi * kFftRow1 + j * 16 + lo];
        // ---- pass B (over n1) + twiddle W256^(n0*k1) -----------------------------------------------------------
        if (!(ABL & 1)) pk_fft16<false>(v);
        QK_STAMP(3)
        QK_LDS_BARRIER();
        QK_STAMP(4)
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[(hi * 16 + k) * kFftRow2 + lo] = (k == 0) ? v[rev16(0)] : abl_cmul<false, ABL>(v[rev16(k)], tb[k]);
        QK_STAMP(5)
        QK_LDS_BARRIER();
        QK_STAMP(6)
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[t * kFftRow2 + j];
        // ---- pass C (over n0), spectrum * Hf, pass C' (over k2) ---------------------------------------------------
        if (!(ABL & 1)) pk_fft16<false>(v);
        v2f y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = abl_cmul<false, ABL>(v[rev16(k)], hf[k]);
        if (!(ABL & 1)) pk_fft16<true>(y);
        QK_STAMP(7)
        QK_LDS_BARRIER();
        QK_STAMP(8)
#pragma unroll
        for (int j = 0; j < 16; j++)
            lds[t * kFftRow2 + j] = (j == 0) ? y[rev16(0)] : abl_cmul<true, ABL>(y[rev16(j)], tb[j]);
        QK_STAMP(9)
        QK_LDS_BARRIER();
        QK_STAMP(10)
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = lds[(hi * 16 + j) * kFftRow2 + lo];
        // ---- pass B' (over k1) --------------------------------------------------------------------------------------
        if (!(ABL & 1)) pk_fft16<true>(v);
        QK_STAMP(11)
        QK_LDS_BARRIER();
        QK_STAMP(12)
#pragma unroll
        for (int j = 0; j < 16; j++) lds[hi * kFftRow1 + j * 16 + lo] = v[rev16(j)];
        QK_STAMP(13)
        QK_LDS_BARRIER();
        QK_STAMP(14)
        v2f e[16];
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = lds[k * kFftRow1 + te];
        // ---- the next segment's samples: requested now, into the columns this wave has just read ---------------------
        const int nb = b + a.nwg;
        landed_in_lds = is_interior(nb);
        if (landed_in_lds) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads above have returned
            request(nb);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = (k == 0) ? e[k] : abl_cmul<true, ABL>(e[k], ta[k]);
        // ---- pass A' (over k0) and store the L valid outputs --------------------------------------------------------
        if (!(ABL & 1)) pk_fft16<true>(v);
        eight_younger = false;
        if (seg0 >= 0 && seg0 + kFftN <= a.nout) {
            // 16-byte stores: the lane with half == 0 stores both columns of its pair for rows 0-7, its partner for rows 8-15
            // (one wave instruction = four 256-byte runs: two of row r, two of row r + 8)
            f4v* __restrict__ o4 = reinterpret_cast<f4v*>(a.out + seg0);   // wave-uniform; 16-byte units
            int pair = (te >> 1) + half * 8 * 128;
            asm volatile("" : "+v"(pair));   // (as above)
            const bool nts = (a.nt & 2) != 0;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const v2f lo_row = v[rev16(r)], hi_row = v[rev16(8 + r)];
                const auto sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo_row.x), __float_as_uint(hi_row.x), false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo_row.y), __float_as_uint(hi_row.y), false, false);
                // now (x,y | z,w) = columns (te & ~1, +1) of row half*8 + r
                if ((half * 8 + r) * 256 + (te & ~1) >= a.ov) {   // ov is even: both elements or neither
                    const f4v o = {__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sx[1]), __uint_as_float(sy[1])};
                    if (ABL & 2) { if (o.x == 1.2345e30f) o4[pair + r * 128] = o; }
                    else if (nts) __builtin_nontemporal_store(o, o4 + pair + r * 128);
                    else o4[pair + r * 128] = o;
                }
            }
            eight_younger = true;   // (rows 8..15 always lie past the overlap: ov <= 2048 on this path, so all eight are issued)
        } else {
            const int first_out = seg0 < 0 ? (int)-seg0 : 0;                                          // output index >= 0
            const int from = a.ov > first_out ? a.ov : first_out;
            const int end_out = a.nout - seg0 < kFftN ? (int)(a.nout - seg0) : kFftN;
            v2f* __restrict__ seg_out = reinterpret_cast<v2f*>(a.out) + seg0;
            int tev = te;
            asm volatile("" : "+v"(tev));   // (as above)
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const int i = n2 * 256 + tev;
                if (i >= from && i < end_out) seg_out[i] = v[rev16(n2)];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STAMPS) {
        if (t == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) a.stamps[(long long)blockIdx.x * 16 + q] = acc[q];
        }
    }
#undef QK_STAMP
}

// Decimating variant, grouped: one workgroup runs the forward half (load [+NCO], passes
// A, B, C, spectrum product, pass C') of DEC consecutive segments, each leaving only its
// 16/DEC wanted values of n0 per lane in an LDS staging area; then ONE full-width inverse
// (passes B', A' on all 256 lanes) finishes the DEC segments together.  Compared with
// pruning inside a single segment (256/DEC active lanes, latency-bound) every pass runs on
// all lanes and the inverse costs 1/DEC per segment.  The next segment's samples are
// prefetched into registers while the current one is in passes B and C.
// LDS: exchange buffer + staging = 2 x 34 KB -> 2 workgroups per CU.
template <int DEC, bool ROT>
__global__ __launch_bounds__(kFftNT, 2) void fir_fft_dec_kernel(const FftArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[2 * kFftLdsElems + 16 * 17];
    float2* stage = lds + kFftLdsElems;      // [DEC*NACT rows][17]
    float2* tbl = lds + 2 * kFftLdsElems;    // pass-B twiddles
    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int H = a.H;
    constexpr int NS = 16 / DEC;
    constexpr int NACT = 256 / DEC;

    if ((int)blockIdx.x == a.nwg) {
        for (int i = t; i < H; i += kFftNT) {
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
                    // the un-rotated form (gain kept) for the next overlap-save call: saves its de-rotation launch
                    if (a.hist_raw_next) a.hist_raw_next[i] = make_float2(v.x * gain, v.y * gain);
                    v = cmulc<false>(v, make_float2((float)p.x * gain, (float)p.y * gain));
                }
            }
            a.hist_next[i] = v;
        }
        return;
    }

    // per-lane constants kept in registers: forward twiddles and the lane's spectrum slice; the
    // inverse-pass twiddles (used once per group) are re-read from the L2-resident table
    float2 ta[16], hf[16];
    // inverse-pass lane w = bb*NACT + n1*NS + s  <->  element e0 = n1*16 + s*DEC of segment bb
    const int wbb = t / NACT, wr = t % NACT;
    const int e0 = (wr / NS) * 16 + (wr % NS) * DEC;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hf[k] = a.Hf[t * 16 + k];
        ta[k] = a.TA[t * 16 + k];
    }
    tbl[(t >> 4) * 17 + (t & 15)] = a.TB[t];
    const float2* tb = tbl + lo * 17;

    float2 pl = make_float2(1.0f, 0.0f);   // exp(j 2pi e0 dphase): the lane's outputs sit at elements e0 + 256 n2
    if (ROT) {
        const double2 p = fx_phasor((unsigned long long)e0 * a.dphase);
        pl = make_float2((float)p.x, (float)p.y);
    }

    auto load_segment = [&](int b, float2 (&v)[16]) {
        const long long seg0 = (long long)b * a.L - a.seg_shift;
        if (b < a.nblocks && seg0 >= 0 && seg0 + kFftN <= a.count) {
            const float2* __restrict__ p = a.in + seg0 + t;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) v[n2] = p[n2 * 256];
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) {
                const long long g = seg0 + n2 * 256 + t;
                float2 x = make_float2(0.0f, 0.0f);
                if (b < a.nblocks) {
                    if (g < 0) { if (g + H >= 0) x = a.hist[g + H]; }   // (ROT: de-rotated by the host side)
                    else if (g < a.count) x = a.in[g];
                }
                v[n2] = x;
            }
        }
    };

    const int ngroups = (a.nblocks + DEC - 1) / DEC;
    // Output NCO: exact fixed-point phase of this lane's segment in the workgroup's FIRST group (one FP64
    // sincos per launch), then an FP64 rotation per group.  (A sincos per group cost 20 spilled VGPRs in
    // the group loop: its temporaries on top of the two 16-value tables.)
    double2 pbg = make_double2(1.0, 0.0);
    if (ROT) pbg = fx_phasor(a.phase0 + (unsigned long long)((long long)((int)blockIdx.x * DEC + wbb) * a.L - a.seg_shift) * a.dphase);
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += a.nwg) {
        const int b0 = grp * DEC;
        // Output NCO (ROT): every value this lane stores is a kept output of segment b0 + wbb, at positions
        // seg0 + e0 + 256 n2 -- 16 rotations per GROUP (an input-side NCO rotates 16 samples per lane and
        // SEGMENT).
        float2 q = make_float2(1.0f, 0.0f);
        if (ROT) {
            q = cmulc<false>(make_float2((float)pbg.x, (float)pbg.y), pl);
            pbg = dcmul(pbg, a.rot_step);      // this workgroup's next group: nwg*DEC segments further
        }
        float2 v[16], vn[16];
        load_segment(b0, v);
#pragma unroll 1
        for (int bb = 0; bb < DEC; bb++) {
            const int b = b0 + bb;
            if (ROT && a.gm1 != 0.0f) {
                // VOLK's magnitude sawtooth stays on the input samples (see fir_fft_kernel)
                const long long seg0 = (long long)b * a.L - a.seg_shift;
                const int gb = (int)((seg0 + t) & 511);
                const float g0 = fmaf((float)gb, a.gm1, 1.0f), g1 = fmaf((float)(gb ^ 256), a.gm1, 1.0f);
                const bool nohist = seg0 >= 0;
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) {
                    const float gg = (n2 & 1) ? g1 : g0;
                    if (nohist || seg0 + n2 * 256 + t >= 0) v[n2] = make_float2(v[n2].x * gg, v[n2].y * gg);
                }
            }
            // ---- pass A + twiddle ----------------------------------------------------------
            fft16<false>(v);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) lds[k * kFftRow1 + t] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], ta[k]);
            if (bb + 1 < DEC) load_segment(b + 1, vn);   // prefetch: lands during passes B and C
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = lds[hi * kFftRow1 + j * 16 + lo];
            // ---- pass B + twiddle ----------------------------------------------------------
            fft16<false>(v);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++)
                lds[(hi * 16 + k) * kFftRow2 + lo] = (k == 0) ? v[rev16(0)] : cmulc<false>(v[rev16(k)], tb[k]);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = lds[t * kFftRow2 + j];
            // ---- pass C, spectrum product, pass C', keep n0 = s*DEC ---------------------------
            fft16<false>(v);
            float2 y[16];
#pragma unroll
            for (int k = 0; k < 16; k++) y[k] = cmulc<false>(v[rev16(k)], hf[k]);
            // Pass C' pruned to the NS = 16/DEC kept outputs n0 = s*DEC:
            //   Y[s*DEC] = sum_{r < NS} e^{+j 2pi r s / NS} * (sum_{k == r mod NS} y[k])
            // (16 complex adds at DEC = 8 instead of the 72 of a full radix-16 butterfly)
            float2 gsum[NS];
#pragma unroll
            for (int r = 0; r < NS; r++) {
                gsum[r] = y[r];
#pragma unroll
                for (int k = r + NS; k < 16; k += NS) gsum[r] = cadd(gsum[r], y[k]);
            }
            if constexpr (NS == 4) fft4<true>(gsum[0], gsum[1], gsum[2], gsum[3]);
            if constexpr (NS == 2) {
                const float2 e = gsum[0], o = gsum[1];
                gsum[0] = cadd(e, o);
                gsum[1] = csub(e, o);
            }
            // staging is only read after the group's last segment (barrier below): no hazard here
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int n0 = s * DEC;
                stage[(bb * NACT + hi * NS + s) * kFftRow2 + lo] = (n0 == 0) ? gsum[0] : cmulc<true>(gsum[s], tb[n0]);
            }
            if (bb + 1 < DEC) {
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) v[n2] = vn[n2];
            }
        }
        __syncthreads();
        // ---- inverse for the whole group: pass B' (lane u = bb*NACT + k0*NS + s, over k1) --------
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = stage[t * kFftRow2 + j];
        fft16<true>(v);
        {   // -> row (bb*NACT + n1*NS + s), column k0, in the exchange buffer (free since pass C)
            const int ubb = t / NACT, ur = t % NACT;
            const int k0 = ur / NS, s = ur % NS;
#pragma unroll
            for (int j = 0; j < 16; j++) lds[(ubb * NACT + j * NS + s) * kFftRow2 + k0] = v[rev16(j)];
        }
        __syncthreads();
        // ---- pass A' (lane w = bb*NACT + n1*NS + s, over k0) and store -----------------------------
        {
            const float2* ta2 = a.TA + e0 * 16;
            asm volatile("" : "+v"(ta2) : "v"(v[0].x));  // opaque: re-read per group, after pass B'
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float2 e = lds[t * kFftRow2 + k];
                v[k] = (k == 0) ? e : cmulc<true>(e, ta2[k]);
            }
        }
        fft16<true>(v);
        const int b = b0 + wbb;
        const long long nb = ((long long)b * a.L - a.ov) / DEC;  // (b*L + i - ov)/DEC at i = 0 (ov, L multiples of DEC)
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) {
            const int i = n2 * 256 + e0;
            const long long n = nb + i / DEC;
            float2 y = v[rev16(n2)];
            if (ROT) y = cmulc<false>(y, (n2 == 0) ? q : cmulc<false>(q, a.wtab[n2]));
            if (b < a.nblocks && i >= a.ov && n < a.nout) a.out[n] = y;
        }
        // next group's pass-A write to `lds` is behind that group's first barrier; its staging
        // writes are behind several more: no extra barrier needed here.
    }
}

int launch_fir_fft(const FftArgs& a, int grid, hipStream_t stream) {
#define QK_FFT(dec, rot) hipLaunchKernelGGL((fir_fft_dec_kernel<dec, rot>), dim3(grid), dim3(kFftNT), 0, stream, a)
    const bool r = a.rot != 0;
    switch (a.dec) {
        case 1:  // FIR, or any-decimation resampler / VFO through the strided store (a.decm)
#if QDSP_HIP_DIAG   // (make DIAG=1: the ablation builds and the in-kernel time stamps behind profiles/r03_ablate_*.txt, r03_stamp_*.txt)
            if (a.dma == 2 && a.abl == 1) hipLaunchKernelGGL((fir_fft_dmapk_kernel<false, 1>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma == 2 && a.abl == 2) hipLaunchKernelGGL((fir_fft_dmapk_kernel<false, 2>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma == 2 && a.abl == 4) hipLaunchKernelGGL((fir_fft_dmapk_kernel<false, 4>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma == 2 && a.abl == 6) hipLaunchKernelGGL((fir_fft_dmapk_kernel<false, 6>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma == 2 && a.abl == 7) hipLaunchKernelGGL((fir_fft_dmapk_kernel<false, 7>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma == 2 && a.stamps) hipLaunchKernelGGL(fir_fft_dmapk_kernel<true>, dim3(grid), dim3(kFftNT), 0, stream, a);
            else
#endif
            if (a.dma == 2) hipLaunchKernelGGL(fir_fft_dmapk_kernel<false>, dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.dma) hipLaunchKernelGGL(fir_fft_dma_kernel, dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (a.real2) hipLaunchKernelGGL((fir_fft_kernel<1, false, true>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else if (r) hipLaunchKernelGGL((fir_fft_kernel<1, true>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else hipLaunchKernelGGL((fir_fft_kernel<1, false>), dim3(grid), dim3(kFftNT), 0, stream, a);
            break;
        case 2:  // half the outputs are kept: the per-segment pruned inverse (128 active lanes) is enough
            if (r) hipLaunchKernelGGL((fir_fft_kernel<2, true>), dim3(grid), dim3(kFftNT), 0, stream, a);
            else hipLaunchKernelGGL((fir_fft_kernel<2, false>), dim3(grid), dim3(kFftNT), 0, stream, a);
            break;
        // grouped: the chip-filling form; per segment (small calls: a 1e6-sample block is 33 groups of 8 segments, worked
        // through serially by 33 workgroups in 27 us -- as 260 independent segments it takes 9 us)
#define QK_SEG(dec, rot) hipLaunchKernelGGL((fir_fft_kernel<dec, rot>), dim3(grid), dim3(kFftNT), 0, stream, a)
        case 4: if (a.grouped) { if (r) QK_FFT(4, true); else QK_FFT(4, false); } else { if (r) QK_SEG(4, true); else QK_SEG(4, false); } break;
        case 8: if (a.grouped) { if (r) QK_FFT(8, true); else QK_FFT(8, false); } else { if (r) QK_SEG(8, true); else QK_SEG(8, false); } break;
        case 16: if (a.grouped) { if (r) QK_FFT(16, true); else QK_FFT(16, false); } else { if (r) QK_SEG(16, true); else QK_SEG(16, false); } break;
#undef QK_SEG
        default: return -1;
    }
#undef QK_FFT
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
