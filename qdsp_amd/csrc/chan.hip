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
//   lane (n' = l>>2, sub = l&3): U[sub + 4i] from T -> radix-16 DFT over i in registers
//            -> * exp(+-j 2pi c0 sub / 64) -> radix-4 across the quad with DPP (no LDS)
//            -> channels c0 + 16 c1 at time n': rot_c(n'), wave transpose (ds_bpermute) and stores; each
//            store instruction writes 4 runs of 16 consecutive n' (128-byte lines), each run by 16
//            neighbouring lanes.  (The first version scattered 64 8-byte pieces per store and was bound
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
constexpr int kChRowT = 68;              // T[n'][mu] row pitch: lanes (n', sub) read 4n' + sub (mod 32): conflict-free
constexpr int kChWaveLds = kChT * kChRowT + 2 * 64;   // float2 elements per wave: T + {A_c, j A_c}[64]

// quad_perm of both lanes of a packed pair.  One 64-bit update_dpp (expanded to two v_mov_b32_dpp):
// with two 32-bit mov_dpp calls hipcc (ROCm 7.2) keeps only the first and feeds it to both halves of the
// consuming v_pk_fma_f32.
template <int CTRL> __device__ __forceinline__ v2f dpp_quad(v2f v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const long long r = __builtin_amdgcn_update_dpp(0ll, __builtin_bit_cast(long long, v), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(v2f, r);
#else
    return v;   // (the host pass only parses this; the type-generic builtin exists for the device target)
#endif
}

// Both lanes of a packed pair fetched from the lane whose byte index is `src` (two ds_bpermute_b32 through a 64-bit
// integer: taken from .x and .y of the vector directly, hipcc permutes .x only and copies it into both halves).
__device__ __forceinline__ v2f bperm_pair(int src, v2f v) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)bits);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(bits >> 32));
    return __builtin_bit_cast(v2f, ((unsigned long long)hi << 32) | lo);
}

// INV: channel c sits at +c/64 turn per sample relative to channel 0, else -c/64.
// M: decimation, 64 (critically sampled) or 8 / 16 / 32 (oversampled by 64/M): the tap window then
// advances M < 64 samples per output, so (a) the 4 rows an output needs are loaded per output (they
// are not the neighbours' rows any more; the few KB a tile touches stay in L1/L2) and (b) the channel
// phase exp(+-j 2pi c j/64) at j = M n' - P + k leaves a root of unity exp(+-j 2pi c M n'/64) per
// output time that the DFT does not cover: its tile part rides in A_c, its in-tile part depends on
// (c0, n') only and is applied with the wave twiddle.
// (M = 8: its 40-value window takes 80 VGPRs; 2 waves/SIMD without spills measured 10 % faster than 3 with a few)
// ABL (diagnostic builds, profiles/r03_chan_tuning.txt): 1 = no global stores, 2 = no DMA (tiles are whatever lies in LDS), 8 = plain instead of non-temporal stores
// QF: 4 = all four tap rows present (193..256 taps: no per-row test in the accumulation), 0 = a.Q rows (run-time)
template <bool INV, int M, int QF = 0, int ABL = 0>
__global__ __launch_bounds__(256, 3) void chan_uniform_kernel(const ChanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const int P = a.P, Q = QF ? QF : a.Q;
    const v2f* __restrict__ in = reinterpret_cast<const v2f*>(a.in);
    const v2f* __restrict__ hist = reinterpret_cast<const v2f*>(a.hist);

    if ((int)blockIdx.x == a.nwg) {
        // history for the next call: the last P samples of hist ++ in
        for (int i = t; i < P; i += 256) {
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
    v2f* ptw = reinterpret_cast<v2f*>(lds + 4 * kChWaveLds);                  // [c0][lane], shared by the four waves
    auto kSlot = [](int c) { return ((c & 15) << 2) | (c >> 4); };
    {
        const int ln = t & 63, nql = ln >> 2, subl = ln & 3;
        const double2 wd = fx_phasor((unsigned long long)(M * nql) * a.dphase0);
        const v2f W = mk2((float)wd.x, (float)wd.y);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c0 = (t >> 6) * 4 + k;
            float2 w = a.tw64[(c0 * (subl + M * nql)) & 63];    // exp(-j 2pi m / 64); conjugated when INV
            if (INV) w.y = -w.y;
            ptw[c0 * 64 + ln] = pk_cmul2<false>(mk2(w.x, w.y), W);
        }
    }
    __syncthreads();                                 // the only workgroup barrier: once per launch

    // (workgroup -> tile in launch order: dealing consecutive tiles to the same XCD, so that neighbours'
    // 3 shared rows meet in one L2, measured 8 % SLOWER -- the 64 output rows are then written at 8
    // distant fronts instead of one)
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv), nwaves = a.nwg * 4;   // (scalar: the tile index feeds SGPR operands)
    const long long tile_pos = (long long)kChT * M;                       // stream positions per tile

    // ---- branch role: lane = staged column p ------------------------------------------------------
    v2f g[4];                                        // complex taps 64q + p
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float2 gq = q < Q ? a.gtaps[64 * q + l] : make_float2(0.0f, 0.0f);
        g[q] = mk2(gq.x, gq.y);
    }
    const int mu = (l - P) & 63;
    // ---- DFT role: lane = (n' = l >> 2, sub = l & 3) ------------------------------------------------
    const int nq = l >> 2, sub = l & 3;
    const v2f* __restrict__ tw = ptw + l;               // entry c0 at tw[64 c0]
    const int c1 = ((sub & 1) << 1) | (sub >> 1);    // radix-4 output this lane keeps (bit-reversed quad index)
    const v2f sA = (sub & 2) ? mk2(-1.0f, -1.0f) : mk2(1.0f, 1.0f), sB = (sub & 1) ? mk2(-1.0f, -1.0f) : mk2(1.0f, 1.0f);
    // between the two radix-2 stages lane 3 takes the -+j twiddle (the other lanes multiply by 1)
    const v2f Mq = sub == 3 ? (INV ? mk2(0.0f, 1.0f) : mk2(0.0f, -1.0f)) : mk2(1.0f, 0.0f);
    // ---- channel role: lane = channel c ----------------------------------------------------------
    double2 corr, corr_step;
    {
        // rot_c at the tile's first output = channel c's own NCO at the window start j0, + kc delta_c, + the
        // +-c P/64 turn the DFT's branch numbering (mu = p - P) left out
        const long long del = a.ddelta[l];                      // delta_c (tiny, signed)
        const long long sc = INV ? (long long)l : -(long long)l;
        const unsigned long long inc = a.dphase0 + (unsigned long long)del + ((unsigned long long)sc << 58);   // == dphase_c
        const long long j0 = (long long)gw * tile_pos - P;
        corr = fx_phasor(a.phase0 + a.dphi[l] + (unsigned long long)j0 * inc + (unsigned long long)(a.kcentre * del) +
                         ((unsigned long long)(sc * P) << 58));
        corr_step = fx_phasor((unsigned long long)(tile_pos * nwaves) * inc);
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
#pragma unroll
        for (int k = 0; k < kStChunks; k++)
            if (!(ABL & 2)) dma16_to_lds(src + 128 * k, voff, lds_T + (unsigned)(128 * k) * 8u);
    };
    bool requested = false, sixteen_younger = false;
    if (kStaged && gw < a.ntiles && dma_ok(gw)) {
        request(gw);
        requested = true;
    }
    for (int wt = gw; wt < a.ntiles; wt += nwaves) {
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
                    if (sixteen_younger) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the DMA is older than the last tile's 16 stores
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
            requested = dma_ok(wt + nwaves);
            if (requested) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                request(wt + nwaves);
            }
        }
        pk_fft16<INV>(U);                            // over i -> group c0 at U[rev16(c0)]
        const long long nn = n0 + nq;
        const long long j = nn * M - P + a.kcentre;                  // window-centre position of this output
        const float jm = (float)(int)(j & 511);
        const float fl = (float)nq;
        // Stores: in the DFT layout the four lanes of a quad hold four different channels (rows 16 apart in memory),
        // so one store instruction touched 64 separate 32-byte pieces and the L1 tag pipeline, not HBM, set the pace
        // (PMC: TCP busy for the whole kernel, 64 cache accesses per store).  One ds_bpermute pair transposes the wave
        // to lane = 16 b + n': 16 neighbouring lanes then write one whole 128-byte line of channel c0 + 16 c1(b).
        const int tsrc = (((l & 15) << 2) | (l >> 4)) << 2;          // byte index of the source lane 4 n' + b
        const int c1s = (((l >> 4) & 1) << 1) | (l >> 5);              // c1 of the source lane's sub = b
        v2f* __restrict__ o = reinterpret_cast<v2f*>(a.out) + n0 + (l & 15) + (size_t)(16 * c1s) * a.out_stride;
        const bool live = n0 + (l & 15) < a.nout;
        // (two copies of the loop: in the full-tile one nothing depends on `live`, so the compiler keeps
        // it one basic block and overlaps the table reads / DPP hazards of neighbouring groups)
        // `quad`: keep the -ang^2/2 term of B_c(n') = exp(j n' theta_c) (host: only when 15 max|theta_c| > 1e-4: float-rounded
        // uniform plans have theta ~ 1e-7 and the term is < 1e-12)
        auto finish = [&](auto guarded, auto quad) {
#pragma unroll
            for (int c0 = 0; c0 < 16; c0++) {
                v2f z = U[rev16(c0)];
                z = pk_cmul2<false>(z, tw[64 * c0]);
                // stage A: pairs (sub, sub^2), then lane 3's -+j twiddle ; stage B: pairs (sub, sub^1)
                v2f ta = pk_fma(z, sA, dpp_quad<0x4E>(z));
                ta = pk_cmul2<false>(ta, Mq);
                const v2f y = pk_fma(ta, sB, dpp_quad<0xB1>(ta));
                // A_c(tile), B_c(n') and VOLK's magnitude sawtooth
                const float4 A = tab[4 * c0 + c1];
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
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(r) : "v"(v), "v"(t2));
                }
                const v2f rt = bperm_pair(tsrc, r);
                if (ABL & 1) { if (rt.x == 1.2345e30f) o[(size_t)c0 * a.out_stride] = rt; }
                else if (ABL & 8) { if (!decltype(guarded)::value || live) o[(size_t)c0 * a.out_stride] = rt; }
                // results are written once and never read here: non-temporal stores (round 3, interleaved A/B on one box: M = 64
                // 0.4068 -> 0.3846 ms per 2^27 samples, M = 8 2.424 -> 2.394; profiles/r03_chan_tuning.txt)
                else if (!decltype(guarded)::value || live) __builtin_nontemporal_store(rt, o + (size_t)c0 * a.out_stride);
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

size_t chan_uniform_lds_bytes() { return (size_t)(4 * kChWaveLds + 16 * 64) * sizeof(float2); }   // + the [c0][lane] twiddles

int launch_chan_uniform(const ChanArgs& a, int grid, hipStream_t stream) {
    const size_t lds_bytes = chan_uniform_lds_bytes();
#define QK_CHAN(m)                                                                                                     \
    case m:                                                                                                            \
        if (a.Q == 4) {                                                                                                \
            if (a.inv) hipLaunchKernelGGL((chan_uniform_kernel<true, m, 4>), dim3(grid), dim3(256), lds_bytes, stream, a);   \
            else hipLaunchKernelGGL((chan_uniform_kernel<false, m, 4>), dim3(grid), dim3(256), lds_bytes, stream, a);        \
        } else {                                                                                                       \
            if (a.inv) hipLaunchKernelGGL((chan_uniform_kernel<true, m, 0>), dim3(grid), dim3(256), lds_bytes, stream, a);   \
            else hipLaunchKernelGGL((chan_uniform_kernel<false, m, 0>), dim3(grid), dim3(256), lds_bytes, stream, a);        \
        }                                                                                                              \
        break;
#if QDSP_HIP_DIAG   // (make DIAG=1: the ablation builds behind profiles/r03_chan_tuning.txt)
    if (a.abl && a.M == 8 && !a.inv && a.Q == 4) {
        if (a.abl == 1) hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, 1>), dim3(grid), dim3(256), lds_bytes, stream, a);
        else if (a.abl == 2) hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, 2>), dim3(grid), dim3(256), lds_bytes, stream, a);
        else if (a.abl == 8) hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, 8>), dim3(grid), dim3(256), lds_bytes, stream, a);
        else hipLaunchKernelGGL((chan_uniform_kernel<false, 8, 4, 3>), dim3(grid), dim3(256), lds_bytes, stream, a);
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
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
