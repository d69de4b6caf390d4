// pfb_dec.hip -- polyphase overlap-save decimate-by-8 kernel, one wave per segment (design notes: pfb_dec.hip.h).
// Own translation unit, built with -fno-slp-vectorize like fft_fir.hip (scalar f32 butterflies, no v_pk shuffles).
#include "pfb_dec.hip.h"
#include "cfft.hip.h"

namespace qk {

namespace {

// cos / sin of 2 pi n / 64 (forward twiddle W64^n = (c, -s))
__device__ constexpr float kCos64[64] = {
    1.f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f,
    0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.f, -0.0980171403f, -0.195090322f, -0.290284677f,
    -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f,
    -0.956940336f, -0.98078528f, -0.995184727f, -1.f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f,
    -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f,
    0.f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f,
    0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f};
__device__ constexpr float kSin64[64] = {
    0.f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f,
    0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f, 1.f, 0.995184727f, 0.98078528f, 0.956940336f,
    0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f,
    0.290284677f, 0.195090322f, 0.0980171403f, 0.f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f,
    -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f,
    -1.f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f,
    -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f};

__host__ __device__ constexpr int rev8(int k) { return ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1); }

// In-register 8-point DFT over v[0], v[S], ... v[7S] (decimation in frequency).  X[k] is left at v[rev8(k) * S].
template <bool INV, int S> __device__ __forceinline__ void fft8(float2* v) {
    constexpr float r = 0.70710678118654752f;
    const float2 a0 = cadd(v[0], v[4 * S]), b0 = csub(v[0], v[4 * S]);
    const float2 a1 = cadd(v[S], v[5 * S]), d1 = csub(v[S], v[5 * S]);
    const float2 a2 = cadd(v[2 * S], v[6 * S]), d2 = csub(v[2 * S], v[6 * S]);
    const float2 a3 = cadd(v[3 * S], v[7 * S]), d3 = csub(v[3 * S], v[7 * S]);
    // odd half: * W8^n, n = 1, 2, 3 (forward W8 = exp(-j 2pi/8); inverse: conjugate)
    const float2 b1 = INV ? make_float2((d1.x - d1.y) * r, (d1.x + d1.y) * r) : make_float2((d1.x + d1.y) * r, (d1.y - d1.x) * r);
    const float2 b2 = mulj<INV>(d2);
    const float2 b3 = INV ? make_float2((-d3.x - d3.y) * r, (d3.x - d3.y) * r) : make_float2((d3.y - d3.x) * r, (-d3.x - d3.y) * r);
    // 4-point DFTs of (a0..a3) -> X[0,2,4,6] and of (b0..b3) -> X[1,3,5,7]
    const float2 c0 = cadd(a0, a2), e0 = csub(a0, a2), c1 = cadd(a1, a3), e1 = mulj<INV>(csub(a1, a3));
    const float2 f0 = cadd(b0, b2), g0 = csub(b0, b2), f1 = cadd(b1, b3), g1 = mulj<INV>(csub(b1, b3));
    v[0] = cadd(c0, c1);          // X0
    v[S] = csub(c0, c1);          // X4
    v[2 * S] = cadd(e0, e1);      // X2
    v[3 * S] = csub(e0, e1);      // X6
    v[4 * S] = cadd(f0, f1);      // X1
    v[5 * S] = csub(f0, f1);      // X5
    v[6 * S] = cadd(g0, g1);      // X3
    v[7 * S] = csub(g0, g1);      // X7
}

// value of `v` in the lane this one is paired with by the DPP control CTRL (all rows, all banks)
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppHalfMirror = 0x141;   // lane l <-> 7 - l inside each group of 8
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]

__device__ __forceinline__ float4 lds_read4(const float2* p) { return *reinterpret_cast<const float4*>(p); }

// Results are written once and never read by this kernel: non-temporal stores (12 % of the traffic, but mixed into the
// read stream they cost it a quarter of its rate -- scripts/micro/stream_read.hip: 64 rows read + 8 rows written per
// wave and segment run 4.9-5.0 TB/s of reads with plain stores, 5.2 with non-temporal ones, 6.3 with none).
typedef float v2f_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(float2* p, float2 v) {
#ifdef PFB_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store((v2f_nt){v.x, v.y}, reinterpret_cast<v2f_nt*>(p));
#endif
}
__device__ __forceinline__ float2 load_stream(const float2* p) {
#ifdef PFB_NT_LOADS
    const v2f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v2f_nt*>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}

}  // namespace

// Segment b (one wave): input positions S0 + [0, 4096), S0 = 8 (b Lo - Q); outputs n' = b Lo + a' - (Q-1) for the
// inverse's elements a' in [Q-1, 512).  Lane l = c + 8 g': column c = l & 7, g' = l >> 3.
//
// PH = 2 (round 3): decimate by FOUR with the same eight column transforms.  y4[n] = sum_k h[k] s[4n - N + k] splits by the
// parity of n into two decimate-by-8 filters over the same eight columns: y4[2n'] has its window end at 8n' - 1 (taps g),
// y4[2n' - 1] at 8n' - 5 (taps g delayed by four samples: g_1 = 0 0 0 0 ++ g).  So the forward side -- loads, eight
// 512-point column transforms -- is done ONCE, the column spectra meet TWO sets of filter spectra (G, G1), and two
// 512-point inverses give the even and the odd outputs: 9 N + 2 (N/8) 9 butterfly stages' worth per N input samples where
// four 1024-point columns + one 1024-point inverse would take 12.5 N.  Eight waves per workgroup share the two tables
// (G1 unpadded, its 16-byte chunks swizzled by the row so that wide reads stay conflict-free); the second phase's column
// sums wait in a wave-private LDS patch (no register for them in the rounds), then take the same inverse.
//
// RD (round 4): REAL data -- PolyphaseResampler<float> with interp 1, decimation 8 / 4.  A real filter is linear over the reals, so TWO segments of the
// real stream ride one set of complex transforms as re / im: pair p = segments (p, p + ceil(nseg / 2)); the loads are two 4-byte loads per row element,
// the stores two 4-byte stores (each stream's own bounds); everything between is the complex kernel.  in / out / hist are float arrays.
template <bool ROT, int PH, bool RD = false>
__device__ __forceinline__ void pfb_body(const PfbArgs& a) {
    static_assert(!(RD && ROT), "the NCO is a complex operator");
    constexpr int NT = kPfbNT * PH, NW = NT / 64;
    __shared__ __attribute__((aligned(16))) float2 sTab[kPfbTableElems + (PH - 1) * kPfbG1Elems];   // G | TW | TI1 | TI2 | EL [| G1]
    __shared__ __attribute__((aligned(16))) float2 sEx[NW][64 * kPfbRow];                          // wave-private exchange buffers
    const int t = threadIdx.x;
    const int H = a.H;

    if ((int)blockIdx.x == a.nwg) {
        // history hand-over (resampling.h:129): last H samples of hist ++ in, in the form the handle keeps them
        if constexpr (RD) {
            const float* inr = reinterpret_cast<const float*>(a.in);
            const float* hk = reinterpret_cast<const float*>(a.hist_keep);
            float* hn = reinterpret_cast<float*>(a.hist_next);
            for (int i = t; i < H; i += NT) {
                const long long g = a.count - H + i;
                hn[i] = g < 0 ? hk[g + H] : inr[g];
            }
            return;
        }
        for (int i = t; i < H; i += NT) {
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
        return;
    }

    // tables -> LDS, once per workgroup (51 KB from L2)
    {
        const float4* src = reinterpret_cast<const float4*>(a.tables);
        float4* dst = reinterpret_cast<float4*>(sTab);
        for (int i = t; i < (kPfbTableElems + (PH - 1) * kPfbG1Elems) / 2; i += NT) dst[i] = src[i];
    }
    __syncthreads();
    const float2* sG = sTab;
    const float2* sTW = sTab + 512 * kPfbRow;
    const float2* sTI1 = sTW + 64 * kPfbRow;
    const float2* sTI2 = sTI1 + 64 * kPfbRow;

    const int l = t & 63, wv = t >> 6;
    float2* E = sEx[wv];
    float2* EY = nullptr;                    // PH = 2: the second phase's column sums, in the layout the inverse reads
    if constexpr (PH == 2) {
        __shared__ __attribute__((aligned(16))) float2 sEy[NW][64 * 9];
        EY = sEy[wv];
    }
    const float2* sG1 = sTab + kPfbTableElems;
    const int sw = (l >> 2) & 3;             // chunk swizzle of G1's rows
    const int c = l & 7, hi3 = l >> 3;       // writer: (column, g'); round reader: (column, mu = kq)
    const bool cb2 = (c & 4) != 0, cb1 = (c & 2) != 0, cb0 = (c & 1) != 0;

    const int nwaves = a.nwg * NW;
    const int wave0 = (int)blockIdx.x * NW + wv;

    float2 el = make_float2(1.0f, 0.0f);     // exp(j 2pi 8 l dphase): this lane's outputs sit at elements l + 64 b1
    double2 pb = make_double2(1.0, 0.0);     // exp(j 2pi ph(first output of the segment)), advanced per segment
    if (ROT) {
        // output n' sits at stream position 8 n' - 1 and gets phase0 + (8 n' - 1) dphase; n' = b Lo - (Q-1) + a'.
        // Lane part from the host's table, segment part from exactly rounded FP64 powers of the per-segment step:
        // no sincos in this kernel (it cost a small call ~6 us at launch)
        el = sTI2[8 * kPfbRow + l];
        pb = a.pb_base;
#pragma unroll
        for (int k = 0; k < 12; k++)
            if ((wave0 >> k) & 1) pb = dcmul(pb, a.seg_pow[k]);
    }

    // Software pipeline: the NEXT segment's rows are requested during the rounds of the current one, each row into
    // the registers a round has just emptied (row 8 rev8(rho) + q' dies in round rho), so the loads have the rest of the
    // rounds and the whole inverse to land and the wave never sits on an empty segment (PMC of the first version: 37 %
    // of wave cycles in s_waitcnt vmcnt at 2 waves per SIMD).  A segment that is not interior (history / zero fill at
    // the ends of the call) is re-read through the checked path when its turn comes; the prefetch for it reads a
    // clamped in-range address and is discarded.
    const long long last_start = a.count - kPfbSeg;       // (the dispatch guarantees count >= 4096)
    auto seg_start = [&](int b) { return 8LL * ((long long)b * a.Lo - a.Q); };
    auto clamped = [&](long long s0) { return s0 < 0 ? 0LL : (s0 > last_start ? last_start : s0); };
    // reduce-scatter over the eight columns (lanes c = 0..7 of each group of 8): lane c ends up with the sum for kg = c.
    // (Round 3 tried the other road -- transpose through the exchange buffer, free between a round's reads and the next round's writes: 8
    // ds_write_b64 + 4 ds_read_b128 + 14 v_add_f32 instead of 14 v_add_f32_dpp + 28 v_cndmask -- and measured 0.2436 against 0.2463 ms
    // per 2^27 samples for the decimator, 0.2560 against 0.2484 for the fused VFO, no change at decimation 4: the ~1300 issue cycles it
    // saves per segment come back as waits on the LDS round trip.  Not kept.)
    auto colsum = [&](const float2 (&pr)[8]) -> float2 {
        float2 s4[4], s2[2];
#pragma unroll
        for (int i = 0; i < 4; i++) {   // partner 7 - c: differs in bit 2; keep the half kg bit 2 == c bit 2
            const float2 keep = cb2 ? pr[4 + i] : pr[i], send = cb2 ? pr[i] : pr[4 + i];
            s4[i] = make_float2(keep.x + dpp<kDppHalfMirror>(send.x), keep.y + dpp<kDppHalfMirror>(send.y));
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {   // partner c ^ 2
            const float2 keep = cb1 ? s4[2 + i] : s4[i], send = cb1 ? s4[i] : s4[2 + i];
            s2[i] = make_float2(keep.x + dpp<kDppXor2>(send.x), keep.y + dpp<kDppXor2>(send.y));
        }
        const float2 keep = cb0 ? s2[1] : s2[0], send = cb0 ? s2[0] : s2[1];      // partner c ^ 1
        return make_float2(keep.x + dpp<kDppXor1>(send.x), keep.y + dpp<kDppXor1>(send.y));
    };
    float2 v[64];
    const float* __restrict__ inr = reinterpret_cast<const float*>(a.in);        // RD: the real stream
    const int nhalf = (a.nseg + 1) >> 1;                                          // RD: pairs; segment b rides with segment b + nhalf
    const int nloop = RD ? nhalf : a.nseg;
    auto segB_of = [&](int b) { return b + nhalf < a.nseg ? b + nhalf : b; };     // (an odd count: the last pair's second member repeats the first, not stored)
    if constexpr (RD) {
        const int b0 = wave0 < nloop ? wave0 : 0;
        const float* __restrict__ pa = inr + clamped(seg_start(b0)) + l;
        const float* __restrict__ pb2 = inr + clamped(seg_start(segB_of(b0))) + l;
#pragma unroll
        for (int r = 0; r < 64; r++) v[r] = make_float2(pa[64 * r], pb2[64 * r]);
    } else {
        const float2* __restrict__ p = a.in + clamped(seg_start(wave0 < a.nseg ? wave0 : 0)) + l;
#pragma unroll
        for (int r = 0; r < 64; r++) v[r] = load_stream(p + 64 * r);
    }
#pragma unroll 1
    for (int b = wave0; b < nloop; b += nwaves) {
        const long long S0 = seg_start(b);
        const int bB = RD ? segB_of(b) : b;
        const long long S0B = seg_start(bB);
        // ---- v[8 j + q'] = seg[512 j + 64 q' + l] (whole 512-byte rows), already requested ---------------------
        if (S0 >= 0 && S0 <= last_start && (!RD || (S0B >= 0 && S0B <= last_start))) {
            if (ROT && a.gm1 != 0.0f) {
                // VOLK's magnitude sawtooth 1 + (g mod 512) gm1 on the INPUT samples: g = S0 + 512 j + 64 q' + l takes
                // eight values per lane and segment (512 j drops out)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float gg = fmaf((float)(int)((S0 + 64 * q + l) & 511), a.gm1, 1.0f);
#pragma unroll
                    for (int j = 0; j < 8; j++) v[8 * j + q] = make_float2(v[8 * j + q].x * gg, v[8 * j + q].y * gg);
                }
            }
        } else if constexpr (RD) {
            const float* hr = reinterpret_cast<const float*>(a.hist);
#pragma unroll
            for (int r = 0; r < 64; r++) {
                const long long gA = S0 + 64 * r + l, gB = S0B + 64 * r + l;
                float xa = 0.0f, xb = 0.0f;
                if (gA < 0) { if (gA + H >= 0) xa = hr[gA + H]; }
                else if (gA < a.count) xa = inr[gA];
                if (gB < 0) { if (gB + H >= 0) xb = hr[gB + H]; }
                else if (gB < a.count) xb = inr[gB];
                v[r] = make_float2(xa, xb);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 64; r++) {
                const long long g = S0 + 64 * r + l;
                float2 x = make_float2(0.0f, 0.0f);
                if (g < 0) { if (g + H >= 0) x = a.hist[g + H]; }       // (ROT: de-rotated by the host side, gain kept)
                else if (g < a.count) {
                    x = a.in[g];
                    if (ROT && a.gm1 != 0.0f) {
                        const float gg = fmaf((float)(int)(g & 511), a.gm1, 1.0f);
                        x = make_float2(x.x * gg, x.y * gg);
                    }
                }
                v[r] = x;
            }
        }
        // where this wave's next segment starts (clamped into the buffer: see above)
        const int bn = b + nwaves < nloop ? b + nwaves : b;
        const float2* __restrict__ pn = a.in + clamped(seg_start(bn)) + l;
        // RD: ONE per-lane pointer (stream A) through the rounds; stream B's is rebuilt in each round from the wave-uniform distance between the two
        // segments (two pointers held across the rounds cost the two-phase body its last two registers; bases in SGPRs with a 32-bit lane offset
        // -- the other way to save them -- run 40 % slower: 95 against 67 us per 2^26 samples at decimation 8)
        const float* __restrict__ pnA = inr + clamped(seg_start(bn)) + l;
        const long long dAB64 = clamped(seg_start(RD ? segB_of(bn) : bn)) - clamped(seg_start(bn));
        unsigned dABlo = __builtin_amdgcn_readfirstlane((unsigned)dAB64), dABhi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)dAB64 >> 32));
        float2 vn[64];
        // ---- forward, column index a = 64 j + 8 q' + g', bin k = k0 + 8 kq + 64 kg ----------------------------
        // pass 1 over j (for every q'): A[k0][q'] at v[8 rev8(k0) + q']
#pragma unroll
        for (int q = 0; q < 8; q++) fft8<false, 8>(v + q);
        // twiddle W64^(q' k0): compile-time constants once the loops are unrolled
#pragma unroll
        for (int k0 = 1; k0 < 8; k0++) {
#pragma unroll
            for (int q = 1; q < 8; q++) {
                const int n = (k0 * q) & 63;
                const float wc = kCos64[n], ws = -kSin64[n];      // W64^n = wc + j ws
                const float2 x = v[8 * rev8(k0) + q];
                v[8 * rev8(k0) + q] = make_float2(fmaf(x.x, wc, -x.y * ws), fmaf(x.x, ws, x.y * wc));
            }
        }
        // pass 2 over q' (for every k0): B[k0][kq] at v[8 rev8(k0) + rev8(kq)]
#pragma unroll
        for (int r = 0; r < 8; r++) fft8<false, 1>(v + 8 * r);

        // ---- eight rounds (k0 = rho): exchange over g', last pass, spectrum product, column sum --------------
        float2 yr[8];   // yr[rho] = Y[rho + 8 mu + 64 c]   (mu = l >> 3, c = l & 7 after the reduce-scatter)
#pragma unroll
        for (int rho = 0; rho < 8; rho++) {
            // writer (g', c): B[rho][kq] -> row (kq, c), column g'
#pragma unroll
            for (int kq = 0; kq < 8; kq++) E[(kq * 8 + c) * kPfbRow + hi3] = v[8 * rev8(rho) + rev8(kq)];
            // the registers of row group rev8(rho) are free now: request the same rows of the next segment
            if constexpr (RD) {
                asm volatile("" : "+s"(dABlo), "+s"(dABhi));    // (opaque per round: pnA + distance is not hoisted into a second pointer)
                const float* __restrict__ pnB = pnA + (long long)(((unsigned long long)dABhi << 32) | dABlo);
#pragma unroll
                for (int q = 0; q < 8; q++) vn[8 * rev8(rho) + q] = make_float2(pnA[64 * (8 * rev8(rho) + q)], pnB[64 * (8 * rev8(rho) + q)]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) vn[8 * rev8(rho) + q] = load_stream(pn + 64 * (8 * rev8(rho) + q));
            }
            __builtin_amdgcn_wave_barrier();
            // reader (c, mu = kq): row l holds the eight g' of bin prefix m = rho + 8 mu
            float2 r8[8], tw[8], gg[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 x = lds_read4(E + l * kPfbRow + 2 * i);
                r8[2 * i] = make_float2(x.x, x.y);
                r8[2 * i + 1] = make_float2(x.z, x.w);
                const float4 w4 = lds_read4(sTW + (rho * 8 + hi3) * kPfbRow + 2 * i);
                tw[2 * i] = make_float2(w4.x, w4.y);
                tw[2 * i + 1] = make_float2(w4.z, w4.w);
                const float4 g4 = lds_read4(sG + (rho * 64 + l) * kPfbRow + 2 * i);
                gg[2 * i] = make_float2(g4.x, g4.y);
                gg[2 * i + 1] = make_float2(g4.z, g4.w);
            }
            __builtin_amdgcn_wave_barrier();
            // twiddle W512^(g' m), pass 3 over g' -> X_c[m + 64 kg] at r8[rev8(kg)]
#pragma unroll
            for (int g = 1; g < 8; g++) r8[g] = cmulc<false>(r8[g], tw[g]);
            fft8<false, 1>(r8);
            float2 pr[8], pr1[PH == 2 ? 8 : 1];   // pr[kg] = G_c[k] X_c[k]  (pr1: against the odd outputs' spectra)
            if constexpr (PH == 2) {
                // (both products before either column sum: the transformed values die here, as they do with one phase -- kept alive across
                // the first reduce-scatter they cost the rounds, the kernel's pressure peak, sixteen registers and 10-33 spills)
                float2 g1[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 g4 = lds_read4(sG1 + (rho * 64 + l) * 8 + 2 * (i ^ sw));
                    g1[2 * i] = make_float2(g4.x, g4.y);
                    g1[2 * i + 1] = make_float2(g4.z, g4.w);
                }
#pragma unroll
                for (int kg = 0; kg < 8; kg++) {
                    pr[kg] = cmulc<false>(r8[rev8(kg)], gg[kg]);
                    pr1[kg] = cmulc<false>(r8[rev8(kg)], g1[kg]);
                }
            } else {
#pragma unroll
                for (int kg = 0; kg < 8; kg++) pr[kg] = cmulc<false>(r8[rev8(kg)], gg[kg]);
            }
            yr[rho] = colsum(pr);
            if constexpr (PH == 2) EY[(rho + 8 * hi3) * 9 + c] = colsum(pr1);
        }

        // ---- inverse 512-point transform of Y: y[a'] = sum_k Y[k] V^(a' k), V = exp(+j 2pi/512) ---------------
        // in: lane m holds z[kg] = Y[m + 64 kg]; out: element a' = l + 64 b1 at z[rev8(b1)]
        auto inverse = [&](float2 (&z)[8]) {
            // pass I1 over kg -> alpha = a' mod 8 at z[rev8(alpha)]; twiddle V512^(m alpha)
            fft8<true, 1>(z);
            float2 zt[8];
            zt[0] = z[0];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 w4 = lds_read4(sTI1 + l * kPfbRow + 2 * i);
                if (i) zt[2 * i] = cmulc<false>(z[rev8(2 * i)], make_float2(w4.x, w4.y));
                zt[2 * i + 1] = cmulc<false>(z[rev8(2 * i + 1)], make_float2(w4.z, w4.w));
            }
            // I2: row alpha, column m  ->  lane (m0 = l & 7, alpha = l >> 3) reads m = m0 + 8 m1
#pragma unroll
            for (int al = 0; al < 8; al++) E[al * 72 + l] = zt[al];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m1 = 0; m1 < 8; m1++) z[m1] = E[hi3 * 72 + c + 8 * m1];
            __builtin_amdgcn_wave_barrier();
            // pass I2 over m1 -> b0 at z[rev8(b0)]; twiddle V64^(m0 b0)   (a' = alpha + 8 b0 + 64 b1)
            fft8<true, 1>(z);
            zt[0] = z[0];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 w4 = lds_read4(sTI2 + c * kPfbRow + 2 * i);
                if (i) zt[2 * i] = cmulc<false>(z[rev8(2 * i)], make_float2(w4.x, w4.y));
                zt[2 * i + 1] = cmulc<false>(z[rev8(2 * i + 1)], make_float2(w4.z, w4.w));
            }
            // I3: row (b0, alpha), column m0  ->  lane alpha + 8 b0 reads its row
#pragma unroll
            for (int b0 = 0; b0 < 8; b0++) E[(8 * b0 + hi3) * 9 + c] = zt[b0];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m0 = 0; m0 < 8; m0++) z[m0] = E[l * 9 + m0];
            __builtin_amdgcn_wave_barrier();
            // pass I3 over m0 -> b1 at z[rev8(b1)]: element a' = l + 64 b1
            fft8<true, 1>(z);
        };
        // ---- store the Lo valid outputs, rotated by the NCO (fused VFO): element a' of phase phi is output PH n' - phi ----
        float2 q = make_float2(1.0f, 0.0f);
        if (ROT) {
            q = cmulc<false>(make_float2((float)pb.x, (float)pb.y), el);
            pb = dcmul(pb, a.rot_step);
        }
        // (32-bit lane offsets from a segment base: per-lane 64-bit output indices get hoisted out of the segment loop and spilled)
        const long long base = PH * ((long long)b * a.Lo - (a.Q - 1));           // output index of element a' = 0, phase 0
        const long long room = a.nout - base;
        const int hi = room > 8192 ? 8192 : (int)room;                            // offsets below this one are inside the call
        const int lo = base > 0 ? -8 : (int)-base;                                // ... and from this one on not before its start
        float2* __restrict__ ob = a.out + base;
        // RD: the second member of the pair has its own output window
        const long long baseB = PH * ((long long)bB * a.Lo - (a.Q - 1));
        const long long roomB = a.nout - baseB;
        const bool haveB = RD && bB != b;                                                // (a repeated segment stores nothing)
        const int hiB = haveB ? (roomB > 8192 ? 8192 : (int)roomB) : 0;
        const int loB = haveB ? (baseB > 0 ? -8 : (int)-baseB) : 0;
        float* __restrict__ orA = reinterpret_cast<float*>(a.out) + base;
        float* __restrict__ orB = reinterpret_cast<float*>(a.out) + baseB;
        auto store_real = [&](const float2 (&z)[8], int phi) {
#pragma unroll
            for (int b1 = 0; b1 < 8; b1++) {
                const int ap = l + 64 * b1;
                const int off = PH * ap - phi;
                const float2 y = z[rev8(b1)];
                if (ap >= a.Q - 1) {
                    // (decimate by 4: the two phases interleave 4-byte pieces -- plain stores, which L2 merges; non-temporal ones do not combine)
                    if constexpr (PH == 1) {
                        if (off >= lo && off < hi) __builtin_nontemporal_store(y.x, orA + off);
                        if (off >= loB && off < hiB) __builtin_nontemporal_store(y.y, orB + off);
                    } else {
                        if (off >= lo && off < hi) orA[off] = y.x;
                        if (off >= loB && off < hiB) orB[off] = y.y;
                    }
                }
            }
        };
        auto store = [&](const float2 (&z)[8], int phi) {
#pragma unroll
            for (int b1 = 0; b1 < 8; b1++) {
                const int ap = l + 64 * b1;
                const int off = PH * ap - phi;
                float2 y = z[rev8(b1)];
                if (ROT) y = cmulc<false>(y, (b1 == 0) ? q : cmulc<false>(q, a.wtab[b1]));
                if (ap >= a.Q - 1 && off >= lo && off < hi) {
                    // (decimate by 4: the two phases interleave 8-byte pieces; written non-temporally those cost 0.414 against 0.313 ms)
                    if (PH == 1) store_nt(ob + off, y);
                    else ob[off] = y;
                }
            }
        };
        float2 z[8];
        // I0: lane (kg = c, mu) holds Y[rho + 8 mu + 64 kg]  ->  lane m reads Y[m + 64 kg], kg = 0..7
#pragma unroll
        for (int rho = 0; rho < 8; rho++) E[(rho + 8 * hi3) * 9 + c] = yr[rho];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kg = 0; kg < 8; kg++) z[kg] = E[l * 9 + kg];
        __builtin_amdgcn_wave_barrier();
        inverse(z);
        if constexpr (RD) {
            store_real(z, 0);
            if constexpr (PH == 2) {
#pragma unroll
                for (int kg = 0; kg < 8; kg++) z[kg] = EY[l * 9 + kg];
                __builtin_amdgcn_wave_barrier();
                inverse(z);
                store_real(z, 1);
            }
        } else if constexpr (PH == 1 || ROT) {
            // (fused VFO at decimation 4: the kernel has no sixteen registers to spare through the second inverse -- pairing the stores as below
            // spilled 11 VGPRs, parking the even outputs in LDS 2 -- so it keeps the phase-by-phase 8-byte stores)
            store(z, 0);
            if constexpr (PH == 2) {
#pragma unroll
                for (int kg = 0; kg < 8; kg++) z[kg] = EY[l * 9 + kg];
                __builtin_amdgcn_wave_barrier();
                inverse(z);
                store(z, 1);
            }
        } else {
            // Decimate by 4 (round 4): the even outputs wait in registers for the odd ones and every lane writes the PAIR (2 a' - 1, 2 a') with one
            // 16-byte store -- consecutive lanes, consecutive 16-byte pieces.  Written phase by phase (round 3) the two sets of 8-byte pieces
            // interleaved in memory and every 64-byte line went out twice: counter traffic x 1.12 of the algorithmic bytes, decim4 0.313 ms.
            float2 y0[8];
#pragma unroll
            for (int b1 = 0; b1 < 8; b1++) y0[b1] = z[rev8(b1)];
#pragma unroll
            for (int kg = 0; kg < 8; kg++) z[kg] = EY[l * 9 + kg];
            __builtin_amdgcn_wave_barrier();
            inverse(z);
            typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));       // (the pair starts on an odd sample: 8-byte aligned)
#pragma unroll
            for (int b1 = 0; b1 < 8; b1++) {
                const int ap = l + 64 * b1;
                const float2 y1 = z[rev8(b1)];
                const bool ok = ap >= a.Q - 1;
                const bool ok0 = ok && 2 * ap >= lo && 2 * ap < hi, ok1 = ok && 2 * ap - 1 >= lo && 2 * ap - 1 < hi;
                if (ok0 && ok1) *reinterpret_cast<f4u*>(ob + (2 * ap - 1)) = (f4u){y1.x, y1.y, y0[b1].x, y0[b1].y};
                else {
                    if (ok1) ob[2 * ap - 1] = y1;
                    if (ok0) ob[2 * ap] = y0[b1];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 64; r++) v[r] = vn[r];
    }
}

template <bool ROT> __global__ __launch_bounds__(kPfbNT, 2) void pfb_dec8_kernel(const PfbArgs a) { pfb_body<ROT, 1>(a); }
template <bool ROT> __global__ __launch_bounds__(2 * kPfbNT, 1) void pfb_dec4_kernel(const PfbArgs a) { pfb_body<ROT, 2>(a); }
__global__ __launch_bounds__(kPfbNT, 2) void pfb_dec8_real_kernel(const PfbArgs a) { pfb_body<false, 1, true>(a); }
__global__ __launch_bounds__(2 * kPfbNT, 1) void pfb_dec4_real_kernel(const PfbArgs a) { pfb_body<false, 2, true>(a); }

int launch_pfb_dec(const PfbArgs& a, hipStream_t stream) {
    if (a.real) {
        if (a.rot) return -1;
        if (a.PH == 2) hipLaunchKernelGGL(pfb_dec4_real_kernel, dim3(a.nwg + 1), dim3(2 * kPfbNT), 0, stream, a);
        else hipLaunchKernelGGL(pfb_dec8_real_kernel, dim3(a.nwg + 1), dim3(kPfbNT), 0, stream, a);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : -(int)e;
    }
    if (a.PH == 2) {
        if (a.rot) hipLaunchKernelGGL((pfb_dec4_kernel<true>), dim3(a.nwg + 1), dim3(2 * kPfbNT), 0, stream, a);
        else hipLaunchKernelGGL((pfb_dec4_kernel<false>), dim3(a.nwg + 1), dim3(2 * kPfbNT), 0, stream, a);
    } else if (a.rot) hipLaunchKernelGGL((pfb_dec8_kernel<true>), dim3(a.nwg + 1), dim3(kPfbNT), 0, stream, a);
    else hipLaunchKernelGGL((pfb_dec8_kernel<false>), dim3(a.nwg + 1), dim3(kPfbNT), 0, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
