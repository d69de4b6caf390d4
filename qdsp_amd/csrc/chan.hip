// chan.hip -- uniform polyphase channelizer: 64 frequency-translating decimators in one pass.
//
// Reference shape: Splitter -> 64 x VFO (src/dsp/routing.h:47-57, src/dsp/vfo.h:19-36), i.e.
// for channel c   y_c[n'] = sum_k h[k] * x[j] * exp(j*phase_c(j)),   j = n'*64 - P + k,
// with phase_c(j) = phi_c + j*dphi_c advancing by that channel's own (rounded-float) phase
// increment (src/dsp/processing.h:20,64).  When the 64 increments are uniformly spaced,
//     dphi_c = dphi_0 +- c/64 turn + delta_c        (delta_c ~ 1e-8 turn: float rounding),
// and the decimation is 64, the sum factors (k = 64q + p, mu = (p - P) mod 64):
//     y_c[n'] = rot_c(n') * sum_mu exp(+-j 2pi c mu / 64) * U[mu],
//     U[mu]   = sum_q g[64q + p] * x[n'*64 - P + 64q + p],      g[k] = h[k] * exp(j k dphi_0),
//     rot_c(n') = exp(j (phi_c + (64n' - P) dphi_0 + (64n' - P + kc) delta_c)),
// a 64-branch polyphase filter with COMPLEX taps (channel 0's mixer folded into the prototype)
// followed by ONE 64-point DFT across the branches for all 64 channels: ~45 FLOP per input
// sample instead of 64 x 134, the input is read once and never rotated.  The only
// approximation is delta_c applied at the centre kc of the tap window instead of per tap
// (error <= 2pi * 1e-8 * ntaps/2 at the window edges, ~1e-6 after tap weighting); VOLK's
// magnitude sawtooth (kernels.hip.h rotate()) is applied the same way per channel.
//
// Waves are independent (no workgroup barrier): one wave = one tile of 16 output times x 64 channels,
// persistent over tiles, 4 waves per workgroup only for launch economy.
//   lane p:  16 branch sums for staged column p straight from global memory (19 rows of 64
//            consecutive samples = 512-byte coalesced loads, sliding 4-row window, complex taps in 8
//            VGPRs) -> wave-private LDS tile T[n'][mu]
//   lane (sub = l>>4, n' = l&15): U[sub + 4i] from T -> radix-16 DFT over i in registers
//            -> * exp(+-j 2pi c0 sub / 64) -> radix-4 across the four 16-lane ROWS of the wave, two groups c0 at a time:
//            v_permlane32_swap pairs rows (sub, sub^2), v_permlane16_swap rows (sub, sub^1) -- a swap hands each lane its
//            butterfly partner AND leaves the results where the stores want them: row r of a result register holds ONE
//            channel c0 + 16 c1 at the 16 consecutive times n' (round 4; rounds 2-3 ran the radix-4 inside the quads with
//            DPP -- lane = 4 n' + sub -- and then transposed the wave with two ds_bpermute per group: 32 LDS operations
//            per tile and a wait in front of every store)
//            -> rot_c(n') and stores; each store instruction writes 4 runs of 16 consecutive n' (128-byte lines), each
//            run by 16 neighbouring lanes.  (The first version scattered 64 8-byte pieces per store and was bound
//            by L2 write requests: 134 M per GiB of output.)
// rot_c(n') = A_c(tile) * W(n' - n0) * B_c(n' - n0): A_c is FP64 state of lane c advanced by one
// complex multiply per tile, W = exp(j 64 m dphi_0) a per-lane constant, B_c = exp(j 64 m delta_c) a
// 2-term FP32 series (<= 2.4e-3 rad over 16 outputs).  History is raw input.
// All complex arithmetic is packed FP32 (cpk.hip.h: v_pk_* does the real and imaginary lane in one
// issue, 1190 -> ~800 VALU instructions per tile).  The kernel runs at the speed of its memory skeleton
// (scripts/micro/write_streams.hip: the same loads and stores without any arithmetic take the same
// 0.43 ms per 2^27 samples; a plain 1 GiB copy takes 0.37-0.41 ms, scripts/micro/copy_bw.hip).
#include "chan.hip.h"
#ifndef QDSP_HIP_DIAG
#define QDSP_HIP_DIAG 0
#endif
#include "cfft.hip.h"
#include "cpk.hip.h"
#include "ldsdma.hip.h"
#include <type_traits>

namespace qk {

constexpr int kChT = 16;                 // output times per wave tile
constexpr int kChRowT = 66;              // T[n'][mu] row pitch in float2: the 32 lanes (sub in {0,1} or {2,3}, n') of one ds_read_b64 group
                                         // read dwords 132 n' + 2 sub + 8 i = 4 n' + 2 sub (mod 64): conflict-free
constexpr int kChWaveLds = kChT * kChRowT + 2 * 64;   // float2 elements per wave: T + {A_c, j A_c}[64]

// Row exchanges of a packed pair of registers (gfx950 v_permlane32_swap / v_permlane16_swap, one per 32-bit half):
//   swap32(a, b): a = [a rows 0 1 | b rows 0 1],  b = [a rows 2 3 | b rows 2 3]
//   swap16(a, b): a = [a row 0, b row 0, a row 2, b row 2],  b = [a row 1, b row 1, a row 3, b row 3]
// quad_perm of both halves of a packed pair
template <int CTRL> __device__ __forceinline__ v2f dpp_pair(v2f v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const long long r = __builtin_amdgcn_update_dpp(0ll, __builtin_bit_cast(long long, v), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(v2f, r);
#else
    return v;
#endif
}
__device__ __forceinline__ void swap32(v2f& a, v2f& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = mk2(__uint_as_float(x[0]), __uint_as_float(y[0]));
    b = mk2(__uint_as_float(x[1]), __uint_as_float(y[1]));
#endif
}
__device__ __forceinline__ void swap16(v2f& a, v2f& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = mk2(__uint_as_float(x[0]), __uint_as_float(y[0]));
    b = mk2(__uint_as_float(x[1]), __uint_as_float(y[1]));
#endif
}

// INV: channel c sits at +c/64 turn per sample relative to channel 0, else -c/64.
// M: decimation, 64 (critically sampled) or 8 / 16 / 32 (oversampled by 64/M): the tap window then
// advances M < 64 samples per output, so (a) the 4 rows an output needs are loaded per output (they
// are not the neighbours' rows any more; the few KB a tile touches stay in L1/L2) and (b) the channel
// phase exp(+-j 2pi c j/64) at j = M n' - P + k leaves a root of unity exp(+-j 2pi c M n'/64) per
// output time that the DFT does not cover: its tile part rides in A_c, its in-tile part depends on
// (c0, n') only and is applied with the wave twiddle.
// (M = 8: its 40-value window takes 80 VGPRs; 2 waves/SIMD without spills measured 10 % faster than 3 with a few)
// ABL (diagnostic builds, profiles/r03_chan_tuning.txt): 1 = no global stores, 2 = no DMA (tiles are whatever lies in LDS), 8 = plain instead of non-temporal stores,
// 4 = DMA requested but never waited for (timing of the landing wait; results wrong), 32 = every DMA read wrapped into a 16 MB window (reads served
// from cache; results wrong).  (16 = the tile's stores in one burst behind the arithmetic was measured and removed: 2.39 -> 2.69 ms)
// QF: 4 = all four tap rows present (193..256 taps: no per-row test in the accumulation), 0 = a.Q rows (run-time)
// WV: waves per workgroup (4).  (16 = one 1024-thread workgroup per CU holding all 160 KiB of LDS, 4 waves per SIMD at 128 VGPRs, was
// measured in round 4: 2.36-2.39 ms against 2.23 at M = 8 -- a fourth wave per SIMD does not help this kernel.)
// ST4 (round 4): 16-byte stores.  Neighbouring lanes (output times n', n' ^ 1) trade one result of each pair of groups, so a lane
// writes two consecutive times of ONE channel: 8 store instructions per tile instead of 16 (and 16 hoisted 64-bit row addresses
// fewer: 158 -> 124 VGPRs at M = 8).  Needs `out` 16-byte aligned and an even row stride (ChanArgs::st4); chan64m8 2.44 -> 2.21 ms.
template <bool INV, int M, int QF = 0, int ABL = 0, int WV = 4, bool ST4 = false>
__global__ __launch_bounds__(64 * WV, M == 64 ? 4 : 3) void chan_uniform_kernel(const ChanArgs a) {
    constexpr int kStores = ST4 ? 8 : 16;            // store instructions of a full tile: what the landing wait of the next tile's DMA leaves in flight
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const int P = a.P, Q = QF ? QF : a.Q;
    const v2f* __restrict__ in = reinterpret_cast<const v2f*>(a.in);
    const v2f* __restrict__ hist = reinterpret_cast<const v2f*>(a.hist);

    if ((int)blockIdx.x == a.nwg) {
        // history for the next call: the last P samples of hist ++ in
        for (int i = t; i < P; i += 64 * WV) {
            const long long g = a.count - P + i;
            a.hist_next[i] = g < 0 ? a.hist[g + P] : a.in[g];
        }
        return;
    }

    const int l = t & 63, wv = t >> 6;
    v2f* T = reinterpret_cast<v2f*>(lds + wv * kChWaveLds);                   // [16][68]
    // Per channel c (slot 4 c0 + c1, kSlot: the four lanes of a quad read c1 = 0..3 of one c0 in the same instruction, and
    // entries 16 apart would share their banks): {A_c of this tile, theta_c, gm1_c} -- one 16-byte read per output.
    float4* tab = reinterpret_cast<float4*>(lds + wv * kChWaveLds + kChT * kChRowT);   // [64]
    // Round 3: ONE twiddle per (c0, lane) instead of three factors: exp(+-j 2pi c0 sub / 64) (the split of the 64-point DFT),
    // exp(+-j 2pi c0 M n'/64) (oversampled plans) -- both 64th roots of unity, so their product is one entry of tw64 -- and
    // W(n') = exp(j M n' dphi_0), common to the four lanes of a quad and therefore free to move in front of the butterfly.
    v2f* ptw = reinterpret_cast<v2f*>(lds + WV * kChWaveLds);                 // [c0][lane], shared by the workgroup's waves
    auto kSlot = [](int c) { return ((c & 15) << 2) | (c >> 4); };
    {
        const int ln = t & 63, nql = ln & 15, subl = ln >> 4;
        const double2 wd = fx_phasor((unsigned long long)(M * nql) * a.dphase0);
        const v2f W = mk2((float)wd.x, (float)wd.y);
#pragma unroll
        for (int k = 0; k < 16 / WV; k++) {
            const int c0 = (t >> 6) * (16 / WV) + k;
            float2 w = a.tw64[(c0 * (subl + M * nql)) & 63];    // exp(-j 2pi m / 64); conjugated when INV
            if (INV) w.y = -w.y;
            ptw[c0 * 64 + ln] = pk_cmul2<false>(mk2(w.x, w.y), W);
        }
    }
    __syncthreads();                                 // the only workgroup barrier: once per launch

    // (workgroup -> tile in launch order: dealing consecutive tiles to the same XCD, so that neighbours'
    // 3 shared rows meet in one L2, measured 8 % SLOWER -- the 64 output rows are then written at 8
    // distant fronts instead of one)
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * WV + wv), nwaves = a.nwg * WV;   // (scalar: the tile index feeds SGPR operands)
    // tile -> wave: round-robin (wave gw takes tiles gw, gw + nwaves, ...).  (Round 4 measured a contiguous run of tiles per wave, so that a
    // wave's successive spans overlap in L1/L2: 2.14 -> 2.49 ms at M = 8; profiles/r04_chan_tuning.txt.)
    const int wt_first = gw, wt_step = nwaves, wt_end = a.ntiles;
    const long long tile_pos = (long long)kChT * M;                       // stream positions per tile

    // ---- branch role: lane = staged column p ------------------------------------------------------
    v2f g[4];                                        // complex taps 64q + p
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float2 gq = q < Q ? a.gtaps[64 * q + l] : make_float2(0.0f, 0.0f);
        g[q] = mk2(gq.x, gq.y);
    }
    const int mu = (l - P) & 63;
    // ---- DFT role: lane = (sub = l >> 4, n' = l & 15) -----------------------------------------------
    const int nq = l & 15, sub = l >> 4;
    const v2f* __restrict__ tw = ptw + l;               // entry c0 at tw[64 c0]
    // After the two row exchanges (finish()) row r of the pair's first result register holds (group ka if r < 2 else kb, c1 = r & 1)
    // and of the second (same group, c1 = 2 + (r & 1)); the odd rows' second radix-2 stage carries the -+j twiddle:
    const int rowq = l >> 4;
    const bool oddrow = (rowq & 1) != 0;
    // ---- channel role: lane = channel c ----------------------------------------------------------
    double2 corr, corr_step;
    {
        // rot_c at the tile's first output = channel c's own NCO at the window start j0, + kc delta_c, + the
        // +-c P/64 turn the DFT's branch numbering (mu = p - P) left out
        const long long del = a.ddelta[l];                      // delta_c (tiny, signed)
        const long long sc = INV ? (long long)l : -(long long)l;
        const unsigned long long inc = a.dphase0 + (unsigned long long)del + ((unsigned long long)sc << 58);   // == dphase_c
        const long long j0 = (long long)wt_first * tile_pos - P;
        corr = fx_phasor(a.phase0 + a.dphi[l] + (unsigned long long)j0 * inc + (unsigned long long)(a.kcentre * del) +
                         ((unsigned long long)(sc * P) << 58));
        corr_step = fx_phasor((unsigned long long)(tile_pos * wt_step) * inc);
    }
    const float theta_c = (float)((double)(a.ddelta[l] * (long long)M) * 3.4061215800865545e-19);   // 2pi / 2^64: exp(j theta_c) per output time
    const float gm1_c = a.gm1[l];

    const v2f* __restrict__ in_or_hist = a.count > 0 ? in : hist;   // any readable address (P >= 1)
    constexpr int kSpan = (kChT - 1) * M + 64 * 4;   // input samples a tile's windows cover (1216 at M = 64)
    // one sample of the first / last tiles' windows: history in front, zeros behind (branch-free)
    auto sample = [&](long long jb, int off) -> v2f {
        const long long gpos = jb + off;
        const bool ok = gpos >= -(long long)P && gpos < a.count;
        const v2f* __restrict__ src = gpos < 0 ? hist + (gpos + P) : in + gpos;
        const v2f v = *(ok ? src : in_or_hist);
        return ok ? v : mk2(0.0f, 0.0f);
    };
    // Oversampled plans: the tile's windows overlap (40 rows of 64 at M = 8 cover only 376 distinct samples), and
    // loading every row from memory made the L1 tag pipeline the bottleneck (PMC: 16 cache accesses per 512-byte
    // load, TCP busy for the whole kernel).  The distinct span is loaded once (kStRows coalesced rows, one tile
    // ahead so the latency hides behind the previous tile's DFT), parked in the T region and the sliding windows
    // are read from LDS.
    constexpr bool kStaged = M != 64;
    // Round 3: the next tile's span is requested by LDS-DMA (ldsdma.hip.h) straight into the T region as soon as this tile's
    // DFT inputs have been read out of it -- no staging registers (the 6 / 8 / 12 prefetched rows cost 12 / 16 / 24 VGPRs and
    // the third wave per SIMD) -- and lands under the finish loop and its sixteen stores.  A request moves 128 samples
    // (lanes x 16 bytes, contiguous in memory and in LDS); it starts on the 16-byte boundary at or below the span
    // (`soff` = 0 or 1 samples of lead) and kStChunks of them cover the span + 1.  Tiles that touch the history or the end
    // of the input are staged by guarded loads when they start (no prefetch: first and last tiles of a call only).
    constexpr int kStChunks = (kSpan + 1 + 127) / 128;
    static_assert(!kStaged || 128 * kStChunks <= kChT * kChRowT, "the staged span fits the T region");
    const unsigned lds_T = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)T);
    const unsigned voff = (unsigned)l * 16u;
    auto span_start = [&](int wt) { return (long long)wt * kChT * M - P; };                 // stream position of the tile's first sample
    auto lead = [&](int wt) { return (int)(((reinterpret_cast<uintptr_t>(in) >> 3) + (unsigned long long)span_start(wt)) & 1); };
    auto dma_ok = [&](int wt) {
        const long long s0 = span_start(wt) - lead(wt);
        return wt < a.ntiles && s0 >= 0 && s0 + 128 * kStChunks <= a.count;
    };
    auto request = [&](int wt) {
        const v2f* src = in + (span_start(wt) - lead(wt));
        if (ABL & 32) src = in + ((span_start(wt) - lead(wt)) & ((1 << 21) - 2));   // (diagnostic: every read inside a 16 MB window -- cache-resident)
#pragma unroll
        for (int k = 0; k < kStChunks; k++)
            if (!(ABL & 2)) dma16_to_lds(src + 128 * k, voff, lds_T + (unsigned)(128 * k) * 8u);
    };
    // The per-lane constants loaded above are waited for HERE.  Left to hipcc the wait lands at their first use -- inside the tile
    // loop, where it can only be `s_waitcnt vmcnt(0)`: every tile then began by waiting for the sixteen stores of the tile before
    // it (round 4: the ISA of rounds 2-3 had exactly that wait at the loop header; chan64m8 2.39 -> see profiles/r04_chan_tuning.txt).
    {
        float touch = theta_c + gm1_c + (float)corr.x + (float)corr_step.x;
#pragma unroll
        for (int q = 0; q < 4; q++) touch += g[q].x + g[q].y;
        asm volatile("" ::"v"(touch));
    }
    bool requested = false, sixteen_younger = false;
    if (kStaged && wt_first < wt_end && dma_ok(wt_first)) {
        request(wt_first);
        requested = true;
    }
    for (int wt = wt_first; wt < wt_end; wt += wt_step) {
        const long long n0 = (long long)wt * kChT;            // first output time of the tile
        const long long jb = n0 * M - P + l;                   // stream position of this lane's column in row 0
        const bool interior = jb - l >= 0 && jb - l + kSpan <= a.count;
        {
            tab[kSlot(l)] = make_float4((float)corr.x, (float)corr.y, theta_c, gm1_c);
            corr = dcmul(corr, corr_step);
        }
        // ---- branch sums: U[n'][mu] = sum_q g[64q + p] x[M n' + 64 q + p] --------------------------------
        // Lane p only ever needs its column at stride M: x_i = x[jb + M i], and output n' uses x_{n' + (64/M) q}.
        // 16 + 3*64/M coalesced 512-byte loads per tile (19 at M = 64, where the rows of neighbouring
        // outputs coincide; 40 at M = 8), a sliding window in registers.
        {
            constexpr int RQ = 64 / M;
            constexpr int NX = kChT + 3 * RQ;
            if constexpr (kStaged) {
                int soff = 0;
                if (requested) {
                    soff = lead(wt);
                    if (ABL & 4) {}                                                             // (diagnostic: timing without the landing wait)
                    else if (sixteen_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");   // the DMA is older than the last tile's stores
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    constexpr int kStRows = (kSpan + 63) / 64;
                    int lv = l;
                    asm volatile("" : "+v"(lv));   // opaque: no per-lane 64-bit addresses hoisted out of the tile loop
#pragma unroll
                    for (int k = 0; k < kStRows; k++) T[64 * k + l] = sample(jb - l + lv, 64 * k);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const v2f* __restrict__ col = T + l + soff;
                // Round 3: the window is no longer held in registers (NX = 40 values = 80 VGPRs at M = 8).  The staged column is
                // walked from its LAST value down; value r feeds the outputs n = r - RQ q, so 3 RQ + 1 accumulators are live at a
                // time, and output n is complete once value r = n has been added: it goes to row n of T right away, IN PLACE --
                // what is still to be read then lies below element M (n - 1) + 64 <= 68 n, the start of row n (the wave runs in
                // lockstep and its LDS operations execute in order).
                v2f acc[kChT];
#pragma unroll
                for (int r = NX - 1; r >= 0; r--) {
                    const v2f xr = col[M * r];
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        const int n = r - RQ * q;
                        if (n >= 0 && n < kChT) {
                            if (q == 3) acc[n] = mk2(0.0f, 0.0f);     // first touch of output n (r = n + 3 RQ)
                            if (q < Q) acc[n] = pk_cmac2(xr, g[q], acc[n]);
                        }
                    }
                    if (r < kChT) T[r * kChRowT + mu] = acc[r];
                }
            } else {
                v2f x[NX];
                if (interior) {
#pragma unroll
                    for (int r = 0; r < NX; r++) x[r] = in[jb + M * r];
                } else {
#pragma unroll
                    for (int r = 0; r < NX; r++) x[r] = sample(jb, M * r);
                }
#pragma unroll
                for (int n = 0; n < kChT; n++) {
                    v2f acc = mk2(0.0f, 0.0f);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < Q) acc = pk_cmac2(x[n + RQ * q], g[q], acc);
                    T[n * kChRowT + mu] = acc;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 64-point DFT over mu = sub + 4i: radix-16 in registers, radix-4 across the quad ----------
        v2f U[16];
#pragma unroll
        for (int i = 0; i < 16; i++) U[i] = T[nq * kChRowT + sub + 4 * i];
        if constexpr (kStaged) {
            // T is free once every lane's reads above have returned: request the next tile's span into it
            requested = wt + wt_step < wt_end && dma_ok(wt + wt_step);
            if (requested) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                request(wt + wt_step);
            }
        }
        pk_fft16<INV>(U);                            // over i -> group c0 at U[rev16(c0)]
        const long long nn = n0 + nq;
        const long long j = nn * M - P + a.kcentre;                  // window-centre position of this output
        const float jm = (float)(int)(j & 511);
        const float fl = (float)nq;
        // Stores: row r of a result register is one channel at the tile's 16 consecutive times, so one store instruction writes
        // four whole 128-byte lines (rounds 2-3 transposed the wave with ds_bpermute to get there; the very first version
        // scattered 64 separate 8-byte pieces per store and the L1 tag pipeline, not HBM, set the pace).
        // Per-lane bases are formed HERE, per tile, from an opaque copy of the lane index; the rows of a pair are then a wave-uniform
        // offset away.  (Left visible, hipcc hoists all sixteen 64-bit row addresses out of the tile loop: 32 VGPRs, the fourth wave.)
        int lop = l;
        asm volatile("" : "+v"(lop));
        const int nq_a = lop & 15, rowq_a = lop >> 4;
        const int grp = rowq >> 1;                                     // 0: the pair's first group (ka), 1: its second (kb)
        v2f* __restrict__ o = reinterpret_cast<v2f*>(a.out) + n0 + nq_a + (size_t)(16 * (rowq_a & 1) + (rowq_a >> 1)) * a.out_stride;
        // ST4: the even lane writes times (n', n' + 1) of the pair's FIRST result's channel, the odd lane (n' - 1, n') of the SECOND's (32 rows on)
        v2f* __restrict__ o4 = o + ((nq_a & 1) ? (size_t)32 * a.out_stride - 1 : (size_t)0);
        const bool live = n0 + nq < a.nout;
        // (two copies of the loop: in the full-tile one nothing depends on `live`, so the compiler keeps
        // it one basic block and overlaps the table reads of neighbouring groups)
        // `quad`: keep the -ang^2/2 term of B_c(n') = exp(j n' theta_c) (host: only when 15 max|theta_c| > 1e-4: float-rounded
        // uniform plans have theta ~ 1e-7 and the term is < 1e-12)
        auto finish = [&](auto guarded, auto quad) {
            // y (the DFT output of this lane's channel) -> rot_c(n') y, stored at row `cs` (channel) of the output
            auto rot_store = [&](v2f y, int slot, size_t row_off) -> v2f {
                // A_c(tile), B_c(n') and VOLK's magnitude sawtooth
                const float4 A = tab[slot];
                const v2f v = pk_cmul2<false>(y, mk2(A.x, A.y));
                v2f r;
                if constexpr (decltype(quad)::value) {
                    const float ang = fl * A.z;
                    const float gain = fmaf(jm, A.w, 1.0f);
                    const float ag = ang * gain;
                    const float brg = fmaf(ang * ang, -0.5f, 1.0f) * gain;
                    r = pk_fma(v.yx, mk2(-ag, ag), v * mk2(brg, brg));
                } else {
                    // v (1 + g1) (1 + j ang) with the second-order terms (ang^2 / 2 < 1e-8 / 2 by the host's test, ang g1 < 1e-8) left out:
                    // v + g1 v + ang (j v) -- one packed multiply for (ang, g1) and two packed FMAs
                    const v2f t2 = mk2(fl, jm) * mk2(A.z, A.w);                      // (ang, g1)
                    r = pk_fma(v, t2.yy, v);
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+&v"(r) : "v"(v), "v"(t2));
                }
                if (ST4 && !decltype(guarded)::value) return r;
                if (ABL & 1) { if (r.x == 1.2345e30f) o[row_off] = r; }
                else if (ABL & 8) { if (!decltype(guarded)::value || live) o[row_off] = r; }
                // results are written once and never read here: non-temporal stores (round 3, interleaved A/B on one box: M = 64
                // 0.4068 -> 0.3846 ms per 2^27 samples, M = 8 2.424 -> 2.394; profiles/r03_chan_tuning.txt)
                else if (!decltype(guarded)::value || live) __builtin_nontemporal_store(r, o + row_off);
                return r;
            };
#pragma unroll
            for (int kp = 0; kp < 8; kp++) {
                const int ka = 2 * kp, kb = 2 * kp + 1;                // the two groups c0 of this pair
                v2f za = pk_cmul2<false>(U[rev16(ka)], tw[64 * ka]);
                v2f zb = pk_cmul2<false>(U[rev16(kb)], tw[64 * kb]);
                // first radix-2 stage, rows (sub, sub ^ 2):  za = [a sub 0 1 | b sub 0 1],  zb = [a sub 2 3 | b sub 2 3]
                swap32(za, zb);
                v2f e = za + zb, d = za - zb;                          // rows: [E_a[0], E_a[1], E_b[0], E_b[1]] and the same of O
                // second stage, rows (s', s' ^ 1):  e = [E_a[0], O_a[0], E_b[0], O_b[0]],  d = [E_a[1], O_a[1], E_b[1], O_b[1]]
                swap16(e, d);
                const v2f dj = pk_mulj<INV>(d);
                const v2f wd = oddrow ? dj : d;                        // odd rows (the O halves): * -+j
                const v2f y0 = e + wd, y1 = e - wd;                    // rows: (a, c1 0), (a, c1 1), (b, 0), (b, 1)  and  (a, 2), (a, 3), (b, 2), (b, 3)
                // this lane's channel: group g = grp ? kb : ka, c1 = (rowq & 1) [+ 2 for y1]; table slot 4 g + c1 (kSlot)
                const int g = grp ? kb : ka;
                const v2f r0 = rot_store(y0, 4 * g + (rowq & 1), (size_t)ka * a.out_stride);
                const v2f r1 = rot_store(y1, 4 * g + 2 + (rowq & 1), (size_t)(ka + 32) * a.out_stride);
                if (ST4 && !decltype(guarded)::value) {
                    // 16-byte stores: neighbouring lanes (times n', n' ^ 1) trade one result each -- the even lane ends up with times
                    // (n', n' + 1) of r0's channel, the odd lane with (n' - 1, n') of r1's -- one store instruction = eight 128-byte runs
                    const bool odd = (nq & 1) != 0;
                    const v2f send = odd ? r0 : r1, own = odd ? r1 : r0;
                    const v2f got = dpp_pair<0xB1>(send);                     // quad_perm [1,0,3,2]: the lane's neighbour n' ^ 1
                    const v2f lo2 = odd ? got : own, hi2 = odd ? own : got;
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((v4f){lo2.x, lo2.y, hi2.x, hi2.y}, reinterpret_cast<v4f*>(o4 + (size_t)ka * a.out_stride));
                }
            }
        };
        if (n0 + kChT <= a.nout) {
            if (a.quad) finish(std::false_type{}, std::true_type{});
            else finish(std::false_type{}, std::false_type{});
            sixteen_younger = true;
        } else {
            finish(std::true_type{}, std::true_type{});
            sixteen_younger = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

size_t chan_uniform_lds_bytes(int waves) { return (size_t)(waves * kChWaveLds + 16 * 64) * sizeof(float2); }   // + the [c0][lane] twiddles

int launch_chan_uniform(const ChanArgs& a, int grid, hipStream_t stream) {
    const size_t lds_bytes = chan_uniform_lds_bytes(a.waves);
#define QK_CHAN2(m, q, st)                                                                                               \
    do {                                                                                                                 \
        if (a.inv) hipLaunchKernelGGL((chan_uniform_kernel<true, m, q, 0, 4, st>), dim3(grid), dim3(256), lds_bytes, stream, a);   \
        else hipLaunchKernelGGL((chan_uniform_kernel<false, m, q, 0, 4, st>), dim3(grid), dim3(256), lds_bytes, stream, a);        \
    } while (0)
#define QK_CHAN(m)                                                                                                     \
    case m:                                                                                                            \
        if (a.Q == 4) {                                                                                                \
            if (a.st4) QK_CHAN2(m, 4, true);                                                                           \
            else QK_CHAN2(m, 4, false);                                                                                \
        } else {                                                                                                       \
            if (a.st4) QK_CHAN2(m, 0, true);                                                                           \
            else QK_CHAN2(m, 0, false);                                                                                \
        }                                                                                                              \
        break;
#if QDSP_HIP_DIAG   // (make DIAG=1: the ablation builds behind profiles/r03_chan_tuning.txt, r04_chan_tuning.txt)
    if (a.abl && a.M == 8 && !a.inv && a.Q == 4) {
        switch (a.abl) {
#define QK_ABL(n) case n: if (a.st4) hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, n, 4, true>), dim3(grid), dim3(256), lds_bytes, stream, a); \
                          else hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, n, 4, false>), dim3(grid), dim3(256), lds_bytes, stream, a); break;
            QK_ABL(1) QK_ABL(2) QK_ABL(3) QK_ABL(4) QK_ABL(8) QK_ABL(32) QK_ABL(33)
#undef QK_ABL
            default: return -1;
        }
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : -(int)e;
    }
    if (a.abl == 8 && a.M == 64 && !a.inv && a.Q == 4) {
        hipLaunchKernelGGL((chan_uniform_kernel<false, 64, 4, 8>), dim3(grid), dim3(256), lds_bytes, stream, a);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : -(int)e;
    }
#endif
    switch (a.M) {
        QK_CHAN(64) QK_CHAN(32) QK_CHAN(16) QK_CHAN(8)
        default: return -1;
    }
#undef QK_CHAN
#undef QK_CHAN2
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
