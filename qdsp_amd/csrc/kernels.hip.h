// kernels.hip.h -- device code of libqdsp_hip.so (gfx950 / CDNA4 only).
//
// Three kernels carry the whole path (DESIGN.md "Kernels"):
//   fir_core_kernel     direct-form FIR / integer decimator (interp == 1), optional fused
//                       NCO rotation while staging.  LDS-staged sliding window, taps from
//                       SGPRs, R register-blocked outputs per lane.
//   resamp_any_kernel   any interp/decim (the reference's general polyphase loop).
//   xlate_kernel        stand-alone NCO mixer (elementwise, HBM-bound).
// plus synth_iq_kernel (measurement input) and small helpers.
//
// Reference semantics each kernel reproduces are cited at the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qk {

template <int CH> struct Smp;
template <> struct Smp<1> {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.0f; }
};
template <> struct Smp<2> {
    using T = float2;
    static __device__ __forceinline__ T zero() { return make_float2(0.0f, 0.0f); }
};

__device__ __forceinline__ void mac(float& acc, float h, float x) { acc = fmaf(h, x, acc); }
// complex sample x real tap: ONE v_pk_fma_f32 (both lanes are IEEE fmas: bit-identical to two fmaf)
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mac(float2& acc, float h, float2 x) {
    const v2f_t r = __builtin_elementwise_fma((v2f_t){h, h}, (v2f_t){x.x, x.y}, (v2f_t){acc.x, acc.y});
    acc = make_float2(r.x, r.y);
}

// ---- NCO ------------------------------------------------------------------------------
// Phase is a 64-bit fixed-point fraction of a turn (2^64 == one full turn), so
// phase0 + n*dphase wraps exactly and never drifts; only the final sincos rounds.
__device__ __forceinline__ double2 phasor_fx(unsigned long long ph) {
    const double t = (double)(ph >> 11) * (1.0 / 9007199254740992.0);  // [0,1) turns
    double s, c;
    sincospi(2.0 * t, &s, &c);
    return make_double2(c, s);
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
// x * phase, FrequencyXlator semantics (src/dsp/processing.h:64): (xr + j xi)(pr + j pi).
// `g` = index of the sample within this call, gm1 = |phase_inc| - 1: VOLK's rotator only
// renormalises its recursive phasor every 512 samples (and at the end of a call), so the
// magnitude of what it multiplies by is |phase_inc|^(g mod 512); to first order that is
// 1 + (g mod 512)*gm1 (gm1 ~ 3e-8).  gm1 = 0 turns the emulation off (ideal NCO).
__device__ __forceinline__ float2 rotate(float2 x, double2 p, long long g, float gm1) {
    const float gain = fmaf((float)(int)(g & 511), gm1, 1.0f);
    const float pr = (float)p.x * gain, pi = (float)p.y * gain;
    return make_float2(fmaf(x.x, pr, -x.y * pi), fmaf(x.x, pi, x.y * pr));
}
__device__ __forceinline__ float rotate(float x, double2, long long, float) { return x; }  // CH==1: unused
// the same with the phasor as an FP32 product p * w (staging batches, stage_tile)
__device__ __forceinline__ float2 rotate_f(float2 x, float2 p, float2 w, long long g, float gm1) {
    const float gain = fmaf((float)(int)(g & 511), gm1, 1.0f);
    const float pr = fmaf(p.x, w.x, -p.y * w.y) * gain, pi = fmaf(p.x, w.y, p.y * w.x) * gain;
    return make_float2(fmaf(x.x, pr, -x.y * pi), fmaf(x.x, pi, x.y * pr));
}
__device__ __forceinline__ float rotate_f(float x, float2, float2, long long, float) { return x; }  // CH==1: unused
__device__ __forceinline__ float2 rot_apply(float2 x, float pr, float pi) { return make_float2(fmaf(x.x, pr, -x.y * pi), fmaf(x.x, pi, x.y * pr)); }
__device__ __forceinline__ float rot_apply(float x, float, float) { return x; }  // CH==1: unused

// ---- element-wise two-input blocks (src/dsp/math.h: Add / Substract / Multiply) -------------
// out = a (+, -, *) b over `n4` float4s (+ a scalar tail); CPLX: * is the complex product of
// interleaved pairs.  Products and sums are separately rounded, as VOLK's generic kernels compute
// them: bit-identical to the C oracle.  (`#pragma clang fp contract(off)` on plain operators is what
// keeps them apart: HIP's __fmul_rn / __fadd_rn are inline operators compiled under the default
// contraction mode and still fuse into FMAs after inlining.)
struct EwArgs {
    const float* a;
    const float* b;
    float* out;
    long long n;     // floats
};
template <int OP, bool CPLX> __device__ __forceinline__ float4 ew4(float4 x, float4 y) {
#pragma clang fp contract(off)
    if (OP == 0) return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    if (OP == 1) return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
    if (!CPLX) return make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    return make_float4(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x, x.z * y.z - x.w * y.w, x.z * y.w + x.w * y.z);
}
template <int OP, bool CPLX>
__global__ __launch_bounds__(256) void ew_kernel(const EwArgs a) {
    const long long n4 = a.n >> 2;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a.a);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(a.b);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(a.out);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) o4[i] = ew4<OP, CPLX>(a4[i], b4[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma clang fp contract(off)
        // tail (< 4 floats; an even count for complex data)
        for (long long i = n4 << 2; i < a.n; i += CPLX ? 2 : 1) {
            if (CPLX && OP == 2) {
                const float xr = a.a[i], xi = a.a[i + 1], yr = a.b[i], yi = a.b[i + 1];
                a.out[i] = xr * yr - xi * yi;
                a.out[i + 1] = xr * yi + xi * yr;
            } else {
                for (int k = 0; k < (CPLX ? 2 : 1); k++) {
                    const float x = a.a[i + k], y = a.b[i + k];
                    a.out[i + k] = OP == 0 ? x + y : OP == 1 ? x - y : x * y;
                }
            }
        }
    }
}

// ---- tile staging shared by the direct-form kernels ------------------------------------------
// hist ++ in (rotated by the NCO if ROT) for stream positions base .. base+U-1 -> put(u, sample).
// Eight independent loads are issued before the first is consumed: written as one load per loop trip
// the compiler leaves them serialised (load, wait, LDS store, next load), and a tile's ~35 loads per
// lane then cost ~35 memory latencies -- several times the arithmetic of the tile.
// The fused NCO needs its exact phasor for lane t of tile `tile`: exp(j*(phase0 + (first + tile*S + t)*dphase)),
// S = samples between tiles.  One FP64 sincos per lane and TILE cost these kernels 30-40 % (a tile is only
// 8-16 staged samples per lane); instead the host keeps three small tables of exactly rounded FP64 phasors per
// (dphase, S, NT) -- T0[b] = exp(j b S dphase), T1[a] = exp(j 256 a S dphase), W[t] = exp(j t dphase) -- and a
// tile's lane phasor is e0 * T1[tile >> 8] * T0[tile & 255] * W[t]: three FP64 complex multiplies.
template <class ARGS> __device__ __forceinline__ double2 tile_phasor(const ARGS& a, int tile, long long base, int t) {
    if (!a.nco_tab) return phasor_fx(a.phase0 + (unsigned long long)(base + t) * a.dphase);
    const double2 blk = cmul(cmul(a.nco_e0, a.nco_tab[256 + (tile >> 8)]), a.nco_tab[tile & 255]);
    return cmul(blk, a.nco_tab[256 + a.nco_na + t]);
}

__device__ __forceinline__ float smp_add(float a, float b) { return a + b; }
__device__ __forceinline__ float2 smp_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

template <int CH, int NT, bool ROT, class ARGS, class PUT>
__device__ __forceinline__ void stage_tile(const typename Smp<CH>::T* __restrict__ in, const typename Smp<CH>::T* __restrict__ hist,
                                           int H, long long count, long long base, int U, const ARGS& a, PUT put, int tile = -1) {
    using T = typename Smp<CH>::T;
    const int t = threadIdx.x;
    // NCO: one exact FP64 phasor per lane and batch of 8 staged samples (advanced by an FP64 rotation of 8*NT
    // samples), times an FP32 table exp(j k NT dphase) inside the batch: an FP64 complex multiply per staged
    // sample made the fused VFO 30-50 % slower than the plain decimator in these memory-bound kernels.
    double2 ph;
    if (ROT) ph = tile >= 0 ? tile_phasor(a, tile, base, t) : phasor_fx(a.phase0 + (unsigned long long)(base + t) * a.dphase);
    constexpr int K = 8;    // (16 in one batch measured no faster at 4-5 workgroups per CU)
    for (int u0 = t; u0 < U; u0 += NT * K) {
        T v[K];
        if (base + u0 - t >= 0 && base + u0 - t + NT * K <= count) {       // block-uniform: the whole batch is plain input
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = in[base + u0 + k * NT];
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const long long g = base + u0 + k * NT;
                T x = Smp<CH>::zero();
                if (g < 0) { if (g + H >= 0) x = hist[g + H]; }
                else if (g < count) x = in[g];
                v[k] = x;
            }
        }
        if (ROT) {
            // VOLK's magnitude sawtooth 1 + (g mod 512)*gm1 takes two values over the batch when NT is a multiple of
            // 256 (g advances by NT per element): fold them into the batch phasor once
            const float2 pf = make_float2((float)ph.x, (float)ph.y);
            ph = cmul(ph, a.rot_8nt);
            const long long gb = base + u0;
            const bool whole = gb - t >= 0 && gb - t + NT * K <= count;
            if (NT % 256 == 0 && whole) {
                const int m0 = (int)(gb & 511);
                const float g0 = fmaf((float)m0, a.gm1, 1.0f), g1 = fmaf((float)((m0 + NT) & 511), a.gm1, 1.0f);
                const float2 p0 = make_float2(pf.x * g0, pf.y * g0), p1 = make_float2(pf.x * g1, pf.y * g1);
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const float2 pb = (NT % 512 == 0 || (k & 1) == 0) ? p0 : p1;
                    const float2 w = a.rot_k[k];
                    const float pr = fmaf(pb.x, w.x, -pb.y * w.y), pi = fmaf(pb.x, w.y, pb.y * w.x);
                    v[k] = rot_apply(v[k], pr, pi);
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const long long g = gb + k * NT;
                    if (g >= 0 && g < count) v[k] = rotate_f(v[k], pf, a.rot_k[k], g, a.gm1);   // history is already rotated
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int u = u0 + k * NT;
            if (u < U) put(u, v[k]);
        }
    }
}

// ---- direct-form core -----------------------------------------------------------------
struct CoreArgs {
    const void* in;       // count samples
    void* out;            // nout samples
    const void* hist;     // H samples: the H inputs preceding in[0]
    void* hist_next;      // H samples: written by the extra block, read by the next call
    const float* taps;    // branch-major: tp[m*Q + q] = h[q*M + m], zero padded
    long long count;      // input samples of this call
    long long nout;       // outputs of this call
    int H;                // history length == delay D of the window start
    int M;                // decimation
    int Q;                // taps per branch = ceil(K / M)
    int nblocks;          // compute blocks; block index nblocks updates the history
    int sb;               // LDS branch stride (elements)
    unsigned long long phase0;  // NCO phase of in[0]            (ROT only)
    unsigned long long dphase;  // NCO phase increment per sample (ROT only)
    double2 rot_nt;             // exp(j*2pi*NT*dphase)           (ROT only)
    double2 rot_8nt;            // exp(j*2pi*8*NT*dphase): one staging batch further  (ROT only)
    float2 rot_k[8];            // exp(j*2pi*k*NT*dphase), k = 0..7, FP32                 (ROT only)
    const double2* nco_tab;     // [T0: 256][T1: nco_na][W: NT] unit phasors (see tile_phasor), nullptr: sincos per tile   (ROT only)
    double2 nco_e0;             // exp(j*(phase0 + first*dphase)), first = stream position staged by lane 0 of tile 0
    int nco_na;
    double2 rot_one;            // exp(j*2pi*dphase)              (ROT only)
    double2 rot_2nt;            // exp(j*2pi*2*NT*dphase)         (ROT only)
    float gm1;                  // |phase_inc| - 1, 0 = ideal NCO  (ROT only)
    int vec;                    // 1: `in` is 16-byte aligned -> float4 staging of interior tiles
};

template <int R> __device__ __forceinline__ int slot(int v) {
    // Lane stride of the window reads is R elements; an even R gets one pad element per R
    // so 32 consecutive lanes fall on distinct banks of ds_read_b64 / ds_read_b32.
    return (R % 2 == 0) ? v + v / R : v;
}

// out[n] = sum_{k < M*Q} h[k] * s[n*M + k - H],  s = hist ++ in      (one tile per block)
//   FIR<T>::run              src/dsp/filter.h:63-67      (M = 1, H = ntaps-1)
//   PolyphaseResampler::run  src/dsp/resampling.h:121-125 with interp == 1 (H = P)
//   + FrequencyXlator::run   src/dsp/processing.h:64 applied to `in` while staging (ROT)
// Written as M interleaved branch filters (k = q*M + m) so every branch is a unit-stride
// sliding window: lane t keeps R accumulators for outputs tR..tR+R-1 and a rotating window
// of R samples; each tap costs one LDS read and R (complex: 2R) FMAs with the tap in an SGPR.
// MT = compile-time decimation (0: take a.M at run time) so the de-interleave is shifts.
template <int CH, int R, int NT, bool ROT, int MT>
__global__ __launch_bounds__(NT) void fir_core_kernel(const CoreArgs a) {
    using T = typename Smp<CH>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int t = threadIdx.x;
    const T* __restrict__ in = static_cast<const T*>(a.in);
    const T* __restrict__ hist = static_cast<const T*>(a.hist);
    const int H = a.H, M = MT ? MT : a.M, Q = a.Q;

    if ((int)blockIdx.x == a.nblocks) {
        // History hand-over (filter.h:71 / resampling.h:129): the last H samples of
        // hist ++ in, into the *other* buffer so block 0 of this launch can still read.
        T* __restrict__ hn = static_cast<T*>(a.hist_next);
        for (int i = t; i < H; i += NT) {
            const long long g = a.count - H + i;
            T v;
            if (g < 0) {
                v = hist[g + H];
            } else {
                v = in[g];
                if (ROT) v = rotate(v, phasor_fx(a.phase0 + (unsigned long long)g * a.dphase), g, a.gm1);
            }
            hn[i] = v;
        }
        return;
    }

    constexpr int TILE = NT * R;
    const long long n0 = (long long)blockIdx.x * TILE;
    const long long base = n0 * M - H;  // sample index (relative to in[0]) of tile element 0
    const int V = TILE + Q;             // elements staged per branch
    const int U = V * M;

    // ---- stage hist ++ in (rotated if ROT) into LDS, de-interleaved by branch -----------
    auto put = [&](int u, T v) {
        int m, vv;
        if (M == 1) { m = 0; vv = u; }
        else { vv = u / M; m = u - vv * M; }
        lds[m * a.sb + slot<R>(vv)] = v;
    };
    bool staged = false;
    if constexpr (CH == 2) {
        // Interior tile (block-uniform test): no history, no end of stream -> unchecked
        // 16-byte loads, two samples per lane, from the even sample at or below `base`.
        if (a.vec && base >= 0 && base + U + 2 <= a.count) {
            const int pre = (int)(base & 1);
            const float4* __restrict__ in4 = reinterpret_cast<const float4*>(a.in) + ((base - pre) >> 1);
            const int npairs = (U + pre + 1) >> 1;
            double2 ph;
            if (ROT) ph = phasor_fx(a.phase0 + (unsigned long long)(base - pre + 2 * t) * a.dphase);
#pragma unroll 4
            for (int j = t; j < npairs; j += NT) {
                const float4 v = in4[j];
                float2 s0 = make_float2(v.x, v.y), s1 = make_float2(v.z, v.w);
                const int u0 = 2 * j - pre;
                if (ROT) {
                    const long long g0 = base + u0;
                    s0 = rotate(s0, ph, g0, a.gm1);
                    s1 = rotate(s1, cmul(ph, a.rot_one), g0 + 1, a.gm1);
                    ph = cmul(ph, a.rot_2nt);
                }
                if (u0 >= 0) put(u0, s0);
                if (u0 + 1 < U) put(u0 + 1, s1);
            }
            staged = true;
        }
    }
    if (!staged) stage_tile<CH, NT, ROT>(in, hist, H, a.count, base, U, a, put, (int)blockIdx.x);
    __syncthreads();

    // ---- sliding-window dot products ------------------------------------------------------
    T acc[R];
#pragma unroll
    for (int p = 0; p < R; p++) acc[p] = Smp<CH>::zero();

    for (int m = 0; m < M; m++) {
        const T* B = lds + m * a.sb + slot<R>(t * R);  // element tR of branch m
        const float* __restrict__ hp = a.taps + m * Q;
        T win[R];
#pragma unroll
        for (int p = 0; p < R; p++) win[p] = B[p];
        int q0 = 0;
        for (; q0 + R <= Q; q0 += R) {
            const T* Bn = B + slot<R>(q0 + R);  // q0 is a multiple of R
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float h = hp[q0 + j];
#pragma unroll
                for (int p = 0; p < R; p++) mac(acc[p], h, win[(p + j) % R]);
                win[j] = Bn[j];
            }
        }
        if (q0 < Q) {
            const T* Bn = B + slot<R>(q0 + R);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (q0 + j < Q) {
                    const float h = hp[q0 + j];
#pragma unroll
                    for (int p = 0; p < R; p++) mac(acc[p], h, win[(p + j) % R]);
                    win[j] = Bn[j];
                }
            }
        }
    }

    // ---- store --------------------------------------------------------------------------
    T* __restrict__ out = static_cast<T*>(a.out);
    const long long n = n0 + (long long)t * R;
    if (n + R <= a.nout) {
#pragma unroll
        for (int p = 0; p < R; p++) out[n + p] = acc[p];
    } else {
#pragma unroll
        for (int p = 0; p < R; p++)
            if (n + p < a.nout) out[n + p] = acc[p];
    }
}

// ---- general interp/decim -------------------------------------------------------------
struct AnyArgs {
    const void* in;
    void* out;
    const void* hist;      // P samples
    void* hist_next;
    const float* phases;   // [L][P] as buildTapPhases lays them out (resampling.h:137-166)
    long long count, nout;
    int L, M, P;
    int tile;              // outputs per tile
    int nblocks;           // tiles
    int nwg;               // persistent workgroups (grid = nwg + 1: the last hands over history)
    int xcd_tiles;         // > 0: tiles per XCD -- workgroup b (on XCD b & 7) walks the contiguous range [xcd * xcd_tiles, +xcd_tiles)
                           // so that neighbouring tiles, which share a window of P samples, meet in ONE XCD's L2; 0: tile = b + k nwg
    int step_d, step_p;    // (NT*M) / L and (NT*M) % L
    unsigned pad_inv;      // PAD: ceil(2^32 / M), u / M == (u * pad_inv) >> 32 for every staged index u
    int ks_lanes;          // 0, or lanes per output S = NT / tile (interp 1, tile < NT): each takes ks_chunk taps
    int ks_shift;          // log2(tile)
    int ks_chunk;          // taps per lane, a multiple of 4
    int ks_red;            // element offset of the NT partial sums behind the staged samples
    int Pp;                // LT: row pitch of the phase table in LDS (floats; Pp/4 odd)
    int tap_bytes;         // LT: bytes of LDS the table takes (multiple of 16)
    unsigned long long phase0, dphase;
    double2 rot_nt;        // exp(j*2pi*NT*dphase)   (ROT only)
    double2 rot_8nt;            // exp(j*2pi*8*NT*dphase): one staging batch further  (ROT only)
    float2 rot_k[8];            // exp(j*2pi*k*NT*dphase), k = 0..7, FP32                 (ROT only)
    const double2* nco_tab;     // [T0: 256][T1: nco_na][W: NT] unit phasors (see tile_phasor), nullptr: sincos per tile   (ROT only)
    double2 nco_e0;             // exp(j*(phase0 + first*dphase)), first = stream position staged by lane 0 of tile 0
    int nco_na;
    float gm1;
};

// PolyphaseResampler<T>::run for any interp L / decim M (src/dsp/resampling.h:121-125):
//   y[n] = sum_t phases[(n*M) % L][t] * s[(n*M)/L - P + t],   s = hist ++ in
// One output per lane per pass; the tile's input span is staged once in LDS.  Workgroups are persistent
// over tiles; LT: the whole [L][P] phase table sits in LDS next to the samples (rows padded to an odd
// number of 16-byte units: every lane reads ITS phase's row, and the rows of neighbouring outputs are
// M % L apart), read four taps at a time -- fetching one tap per MAC per lane from memory kept this
// kernel at 0.49 ms per 2^26 samples for 147/160 (44.1 <-> 48 kHz).
// PAD (interp 1, decimation a multiple of 4): neighbouring lanes' windows start M samples apart, which puts them
// on the same few LDS banks (M = 32: all 64 lanes on one, 0.48 ms per 2^26 samples against 0.13 at M = 25).  The
// tile is then staged with one pad element per M samples (index u -> u + u / M): the lane pitch M + 1 is odd,
// and window element k sits at k + k / M from the lane's start (a wave-uniform offset: rows of M).
// Tap split (interp 1, large decimations): LDS holds tile * M samples, so from M ~ 32 on a tile has fewer outputs
// than the workgroup has lanes (M = 100: 64) and most lanes idled through the P-tap loop.  S = NT / tile lanes
// then share an output: lane t works on output t % tile, taps [j, j + 1) * ks_chunk with j = t / tile (whole
// waves or half-waves per j: the window pattern across lanes is unchanged), partial sums meet in LDS and are
// added in the order j = 0 .. S-1.  (Taps of a wave-uniform chunk through scalar loads instead of LDS broadcast
// reads -- a third of the kernel's LDS traffic -- measured SLOWER, 0.44 vs 0.35 ms per 2^27 samples at M = 50,
// 401 taps: SMEM and LDS share lgkmcnt, so every group of four MACs waits for both.  The padded layout for
// M = 2 mod 4 in this form -- pairs instead of groups of four inside a row -- also lost: 0.41 vs 0.35 ms at M = 50;
// its 2-way conflicts cost less than the row bookkeeping.)
template <int CH, int NT, bool ROT, bool LT, bool PAD>
__device__ __forceinline__ void resamp_any_body(const AnyArgs& a) {
    using T = typename Smp<CH>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tl = reinterpret_cast<float*>(smem_raw);                       // LT: [L][a.Pp]
    T* lds = reinterpret_cast<T*>(smem_raw + (LT ? (size_t)a.tap_bytes : 0));
    const int t = threadIdx.x;
    const T* __restrict__ in = static_cast<const T*>(a.in);
    const T* __restrict__ hist = static_cast<const T*>(a.hist);
    const int P = a.P;

    if ((int)blockIdx.x == a.nwg) {
        T* __restrict__ hn = static_cast<T*>(a.hist_next);
        for (int i = t; i < P; i += NT) {
            const long long g = a.count - P + i;
            T v;
            if (g < 0) {
                v = hist[g + P];
            } else {
                v = in[g];
                if (ROT) v = rotate(v, phasor_fx(a.phase0 + (unsigned long long)g * a.dphase), g, a.gm1);
            }
            hn[i] = v;
        }
        return;
    }

    if (LT) {
        for (int i = t; i < a.L * P; i += NT) {
            const int ph = i / P;
            tl[ph * a.Pp + (i - ph * P)] = a.phases[i];
        }
    }
    T* __restrict__ out = static_cast<T*>(a.out);
    // Tiles overlap by the P-sample window (12 % of a 3200-sample tile at the VFO's 401 taps / decimate by 50).  Dealt
    // round-robin, neighbours run on different XCDs (workgroup b sits on XCD b & 7) and each fetches the shared
    // window from memory (PMC round 1: 1.28x the algorithmic bytes); with a contiguous range per XCD the second
    // reader finds it in that XCD's L2 (round 2: 1.009x, profiles/r02_vfo50_summary.json).
    const int xcd = a.xcd_tiles ? (int)(blockIdx.x & 7) : 0;
    const int t_first = a.xcd_tiles ? xcd * a.xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_step = a.xcd_tiles ? (a.nwg >> 3) : a.nwg;
    const int t_end = a.xcd_tiles ? ((xcd + 1) * a.xcd_tiles < a.nblocks ? (xcd + 1) * a.xcd_tiles : a.nblocks) : a.nblocks;
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const long long n0 = (long long)tile * a.tile;
        long long n1 = n0 + a.tile;
        if (n1 > a.nout) n1 = a.nout;
        const long long lo = (n0 * a.M) / a.L - P;           // first staged sample
        const long long hi = ((n1 - 1) * a.M) / a.L;         // one past the last needed sample
        const int span = (int)(hi - lo);
        __syncthreads();                                     // the previous tile's reads are done (and `tl` is written)
        stage_tile<CH, NT, ROT>(in, hist, P, a.count, lo, span, a, [&](int u, T v) {
            lds[PAD ? u + (int)(((unsigned long long)(unsigned)u * a.pad_inv) >> 32) : u] = v;
        });
        __syncthreads();
        // (n*M) / L and % L: one 64-bit division for the lane's first output of the tile, then n += NT moves
        // them by (NT*M) / L and % L with a carry
        long long d;
        int phase;
        {
            const long long i = (n0 + t) * a.M;
            d = i / a.L;
            phase = (int)(i - d * a.L);
        }
        if (a.ks_lanes) {
            const int o = t & (a.tile - 1), j = t >> a.ks_shift;
            const int kb = j * a.ks_chunk;
            const int ke = (kb + a.ks_chunk < P) ? kb + a.ks_chunk : P;
            const float* hp = LT ? tl : a.phases;
            T acc = Smp<CH>::zero();
            if (n0 + o < n1 && kb < ke) {
                const T* w = lds + o * a.M + (PAD ? o : 0);
                int k = kb;
                if (PAD) {
                    // element k sits at k + k / M; kb and M are multiples of 4: a group of four stays inside a row
                    int row = (int)(((unsigned long long)(unsigned)kb * a.pad_inv) >> 32);
                    int r = kb - row * a.M;
                    for (; k + 4 <= ke; k += 4) {
                        float4 h4;
                        if (LT) h4 = *reinterpret_cast<const float4*>(hp + k);
                        else h4 = make_float4(hp[k], hp[k + 1], hp[k + 2], hp[k + 3]);
                        const T* wk = w + k + row;
                        mac(acc, h4.x, wk[0]);
                        mac(acc, h4.y, wk[1]);
                        mac(acc, h4.z, wk[2]);
                        mac(acc, h4.w, wk[3]);
                        r += 4;
                        if (r == a.M) { r = 0; row++; }
                    }
                    for (; k < ke; k++) {
                        mac(acc, hp[k], w[k + row]);
                        if (++r == a.M) { r = 0; row++; }
                    }
                } else {
                    for (; k + 4 <= ke; k += 4) {
                        float4 h4;
                        if (LT) h4 = *reinterpret_cast<const float4*>(hp + k);
                        else h4 = make_float4(hp[k], hp[k + 1], hp[k + 2], hp[k + 3]);
                        mac(acc, h4.x, w[k]);
                        mac(acc, h4.y, w[k + 1]);
                        mac(acc, h4.z, w[k + 2]);
                        mac(acc, h4.w, w[k + 3]);
                    }
                    for (; k < ke; k++) mac(acc, hp[k], w[k]);
                }
            }
            T* red = lds + a.ks_red;
            red[t] = acc;
            __syncthreads();
            if (t < a.tile && n0 + t < n1) {
                T y = red[t];
                for (int jj = 1; jj < a.ks_lanes; jj++) y = smp_add(y, red[jj * a.tile + t]);
                out[n0 + t] = y;
            }
            continue;   // (the barrier at the top of the tile loop also guards `red`)
        }
        for (long long n = n0 + t; n < n1; n += NT) {
            const T* w = lds + (int)(d - P - lo) + (PAD ? (int)(n - n0) : 0);   // PAD: the start is (n - n0) * M, exactly
            const int phase_n = phase;
            d += a.step_d;
            phase += a.step_p;
            if (phase >= a.L) { phase -= a.L; d += 1; }
            T acc = Smp<CH>::zero();
            if (PAD) {
                // interp 1: one phase.  Rows of M window elements (M a multiple of 4, so the 16-byte tap reads stay
                // aligned) lie M + 1 apart; inside a row the loop is the unpadded one.
                const float* hp = LT ? tl : a.phases;
                const T* wr = w;
                for (int k0 = 0; k0 < P; k0 += a.M, wr += a.M + 1) {
                    const int len = (P - k0 < a.M) ? P - k0 : a.M;
                    const float* hr = hp + k0;
                    int k = 0;
                    for (; k + 4 <= len; k += 4) {
                        float4 h4;
                        if (LT) h4 = *reinterpret_cast<const float4*>(hr + k);
                        else h4 = make_float4(hr[k], hr[k + 1], hr[k + 2], hr[k + 3]);
                        mac(acc, h4.x, wr[k]);
                        mac(acc, h4.y, wr[k + 1]);
                        mac(acc, h4.z, wr[k + 2]);
                        mac(acc, h4.w, wr[k + 3]);
                    }
                    for (; k < len; k++) mac(acc, hr[k], wr[k]);
                }
            } else if (LT) {
                const float* hp = tl + phase_n * a.Pp;
                int k = 0;
                for (; k + 4 <= P; k += 4) {
                    const float4 h4 = *reinterpret_cast<const float4*>(hp + k);
                    mac(acc, h4.x, w[k]);
                    mac(acc, h4.y, w[k + 1]);
                    mac(acc, h4.z, w[k + 2]);
                    mac(acc, h4.w, w[k + 3]);
                }
                for (; k < P; k++) mac(acc, hp[k], w[k]);
            } else {
                const float* __restrict__ hp = a.phases + (size_t)phase_n * P;
                for (int k = 0; k < P; k++) mac(acc, hp[k], w[k]);
            }
            out[n] = acc;
        }
    }
}

template <int CH, int NT, bool ROT, bool LT, bool PAD>
__global__ __launch_bounds__(NT) void resamp_any_kernel(const AnyArgs a) {
    resamp_any_body<CH, NT, ROT, LT, PAD>(a);
}

// The same operator for N channels in ONE launch (Splitter -> N x VFO at arbitrary offsets, src/dsp/routing.h:47-57 +
// src/dsp/vfo.h:19-36: non-uniform qdsp_hip_chan_cf32 plans): blockIdx.y = channel.  All channels share the input, the
// taps and the tile geometry; what differs per channel -- the NCO increment with the staging constants derived from it,
// the two history buffers -- sits in a device table that is rewritten only on a retune, what changes every call (the
// NCO phase at the first sample) travels in the kernel arguments, and the output rows lie out_stride apart.  The
// block's input is read from memory once per channel but from L2 after the first (a 1e6-sample block is 8 MB).
constexpr int kAnyBatchMax = 128;           // channels per launch (phase0[] rides in the kernel arguments: 1 KB)
struct AnyChanConst {
    const void* hist[2];                    // the channel's two history buffers; `cur` selects the one to read
    unsigned long long dphase;
    double2 rot_nt, rot_8nt;
    float2 rot_k[8];
    float gm1;
    int pad_;
};
struct AnyBatchArgs {
    AnyArgs a;                              // shared part (in, taps, geometry); per-channel fields are patched in
    const AnyChanConst* tab;                // [nchan]
    long long out_stride;                   // complex samples between the channels' output rows
    int cur;                                // history parity of this call (all channels flip together)
    int use_ptrs;                           // 1: channel c writes to outs[c] (N separate stream buffers: Splitter -> N x VFO inside a graph)
    unsigned long long phase0[kAnyBatchMax];
    void* outs[kAnyBatchMax];
};
template <int NT, bool LT, bool PAD>
__global__ __launch_bounds__(NT) void resamp_any_batch_kernel(const AnyBatchArgs b) {
    const int ch = blockIdx.y;
    AnyArgs a = b.a;
    const AnyChanConst& c = b.tab[ch];
    a.hist = c.hist[b.cur];
    a.hist_next = const_cast<void*>(c.hist[b.cur ^ 1]);
    a.out = b.use_ptrs ? b.outs[ch] : static_cast<void*>(static_cast<float2*>(b.a.out) + (long long)ch * b.out_stride);
    a.phase0 = b.phase0[ch];
    a.dphase = c.dphase;
    a.rot_nt = c.rot_nt;
    a.rot_8nt = c.rot_8nt;
#pragma unroll
    for (int k = 0; k < 8; k++) a.rot_k[k] = c.rot_k[k];
    a.gm1 = c.gm1;
    a.nco_tab = nullptr;
    resamp_any_body<2, NT, true, LT, PAD>(a);
}

// ---- short-filter integer decimator ---------------------------------------------------------
// y[n'] = sum_k h[k] * s[M n' - P + k],  s = hist ++ in      (PolyphaseResampler<T>::run, interp 1,
// src/dsp/resampling.h:121-125), for the filters an SDR chain is made of: decimate by 2..8 with
// 7..~128 taps.  fir_core_kernel de-interleaves its tile into M branches, which at these sizes costs
// more than the arithmetic (runtime u / M per staged sample, M*(R+Q) window reads for M*Q*R MACs, tiles
// capped by LDS: 0.15-0.29 ms per 2^26 samples, and slower the larger M).  Here the tile is staged as
// it lies in memory (one pad element per M*R so the stride-M*R lane pattern falls on distinct banks),
// lane t owns R consecutive outputs and walks its M*(R-1)+P samples ONCE: sample i feeds output r with
// tap i - M r.  Taps come zero-padded by M*(R-1) on both sides so no tap index is ever tested, in
// chunks of M*R consecutive scalars per output (s_load_dwordx8/x16 into SGPRs).  One tile per workgroup:
// a persistent variant (tiles looped inside, the fused NCO's per-lane FP64 phasor carried from tile to tile
// instead of one sincos per lane and tile) made the NCO free but the kernel itself 1.5-1.8x slower
// (0.13 -> 0.23 ms per 2^26 samples at M = 10): block-level overlap of staging and arithmetic is lost.
// Accumulation is in tap
// order with one FMA per tap, so results equal the k-ordered fmaf chain bit for bit; the first and last
// chunks, where some (sample, output) pairs fall outside the window, test the tap index instead of relying
// on the zero padding, so a NaN/Inf sample stays inside the windows that hold it.
struct WinArgs {
    const void* in;
    void* out;
    const void* hist;      // P samples
    void* hist_next;
    const float* taps;     // zero-padded: hp[k + M*(R-1)] = h[k], length nchunks*M*R + M*(R-1)
    long long count, nout;
    int P;                 // window start / history length: taps per phase for the resampler, ntaps-1 for FIR<T>
    int ntaps;
    int nchunks;           // ceil((M*(R-1) + ntaps) / (M*R))
    int nblocks;
    unsigned long long phase0, dphase;
    double2 rot_nt;        // exp(j*2pi*NT*dphase)   (ROT only)
    double2 rot_8nt;            // exp(j*2pi*8*NT*dphase): one staging batch further  (ROT only)
    float2 rot_k[8];            // exp(j*2pi*k*NT*dphase), k = 0..7, FP32                 (ROT only)
    const double2* nco_tab;     // [T0: 256][T1: nco_na][W: NT] unit phasors (see tile_phasor), nullptr: sincos per tile   (ROT only)
    double2 nco_e0;             // exp(j*(phase0 + first*dphase)), first = stream position staged by lane 0 of tile 0
    int nco_na;
    float gm1;
};

template <int CH, int M, int R, int NT, bool ROT>
__global__ __launch_bounds__(NT) void decim_win_kernel(const WinArgs a) {
    using T = typename Smp<CH>::T;
    constexpr int MR = M * R;                       // samples between the windows of neighbouring lanes
    constexpr int PAD = (MR % 2 == 0) ? 1 : 0;      // lane stride MR + PAD elements must be odd
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int t = threadIdx.x;
    const T* __restrict__ in = static_cast<const T*>(a.in);
    const T* __restrict__ hist = static_cast<const T*>(a.hist);
    const int P = a.P;

    if ((int)blockIdx.x == a.nblocks) {
        T* __restrict__ hn = static_cast<T*>(a.hist_next);
        for (int i = t; i < P; i += NT) {
            const long long g = a.count - P + i;
            T v;
            if (g < 0) {
                v = hist[g + P];
            } else {
                v = in[g];
                if (ROT) v = rotate(v, phasor_fx(a.phase0 + (unsigned long long)g * a.dphase), g, a.gm1);
            }
            hn[i] = v;
        }
        return;
    }

    constexpr int TILE = NT * R;
    const long long n0 = (long long)blockIdx.x * TILE;
    const long long base = n0 * M - P;              // stream position of staged element 0
    const int U = TILE * M + a.nchunks * MR;        // staged span: every chunk any lane reads (zero taps beyond P)
    stage_tile<CH, NT, ROT>(in, hist, P, a.count, base, U, a, [&](int u, T v) { lds[u + (PAD ? u / MR : 0)] = v; }, (int)blockIdx.x);
    __syncthreads();

    T acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = Smp<CH>::zero();
    const T* B = lds + t * (MR + PAD);
    for (int c = 0; c < a.nchunks; c++) {
        const T* Bc = B + c * (MR + PAD);
        const float* __restrict__ hp = a.taps + c * MR;          // hp[u - M r + M (R-1)] pairs sample c*MR + u with output r
        T x[MR];
#pragma unroll
        for (int u = 0; u < MR; u++) x[u] = Bc[u];
        if (c * MR >= M * (R - 1) && (c + 1) * MR <= a.ntaps) {
            // every (sample, output) pair of the chunk meets a real tap
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int u = 0; u < MR; u++) mac(acc[r], hp[u + M * (R - 1 - r)], x[u]);
            }
        } else {
            // first / last chunks: a sample outside an output's window must not reach it even as 0 * x
            // (0 * Inf = NaN): the reference's sum never touches it
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int u = 0; u < MR; u++) {
                    const int k = c * MR + u - M * r;
                    const T xz = (k >= 0 && k < a.ntaps) ? x[u] : Smp<CH>::zero();   // a select, not a branch per FMA
                    mac(acc[r], hp[u + M * (R - 1 - r)], xz);
                }
            }
        }
    }

    T* __restrict__ out = static_cast<T*>(a.out);
    const long long n = n0 + (long long)t * R;
    if (n + R <= a.nout) {
#pragma unroll
        for (int r = 0; r < R; r++) out[n + r] = acc[r];
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (n + r < a.nout) out[n + r] = acc[r];
    }
}

// (Tried for LONG filters too -- 256 taps at decimation 8 is only 128 FLOP per input sample, 0.13 ms of
// FMAs per 2^27 samples: chunks of 8 samples, R tap octets per chunk as s_load_dwordx8.  Measured 0.37 ms
// at R = 4 (LDS bandwidth equals the FMA rate: each staged sample is read by P/(M R) = 8 lanes) and
// 0.56 ms at R = 8 (512 B of LDS per lane: 4 waves per CU), against 0.30 ms for the overlap-save kernel.)

// ---- small-interp rational resampler -----------------------------------------------------
// PolyphaseResampler<T>::run, interp L in {2,3,4,5,10}, decim M <= 8 (src/dsp/resampling.h:121-125):
//   y[n] = sum_t phases[(n*M) % L][t] * s[(n*M)/L - P + t],   s = hist ++ in.
// For n = L*j + c the phase (c*M) % L and the offset e_c = (c*M)/L do not depend on j:
//   y[L*j + c] = sum_k h_c[k] * s[D + M*j + e_c - P + k]
// i.e. L decimate-by-M filters ("sub-filters" c) over the SAME input.  The tile's input is staged once,
// de-interleaved by M as in fir_core_kernel, and every lane runs all L sub-filters for its R consecutive j
// with fir_core's sliding register window -- taps are wave-uniform (SGPRs) and each LDS read feeds R
// FMAs, where resamp_any_kernel fetched a per-lane tap from memory for every MAC (3/2, 189 taps:
// 40 -> 186 Gs/s out).  The lane ends up with R*L consecutive outputs, which go out through LDS so the
// stores are coalesced.  R is odd, so the stride-R lane pattern needs no LDS padding and windows may start
// at any offset.
struct LmArgs {
    const void* in;
    void* out;
    const void* hist;      // P samples
    void* hist_next;
    const float* taps;     // [c][m][q] = h_c[q*M + m], h_c = phases[(c*M) % L], zero padded to M*Q
    const float* taps_t;   // M == 1 only: [q][c] = h_c[q]
    long long count, nout;
    int M, P, Q;           // Q = ceil(P / M)
    int e[10];             // e_c = (c*M) / L
    int sb;                // LDS branch stride (elements)
    int nblocks;
    unsigned long long phase0, dphase;
    double2 rot_nt;        // exp(j*2pi*NT*dphase)   (ROT only)
    double2 rot_8nt;            // exp(j*2pi*8*NT*dphase): one staging batch further  (ROT only)
    float2 rot_k[8];            // exp(j*2pi*k*NT*dphase), k = 0..7, FP32                 (ROT only)
    const double2* nco_tab;     // [T0: 256][T1: nco_na][W: NT] unit phasors (see tile_phasor), nullptr: sincos per tile   (ROT only)
    double2 nco_e0;             // exp(j*(phase0 + first*dphase)), first = stream position staged by lane 0 of tile 0
    int nco_na;
    float gm1;
};

template <int CH, int R, int NT, bool ROT, int L>
__global__ __launch_bounds__(NT) void resamp_lm_kernel(const LmArgs a) {
    static_assert(R % 2 == 1, "odd R: conflict-free stride-R LDS reads without padding");
    using T = typename Smp<CH>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int t = threadIdx.x;
    const T* __restrict__ in = static_cast<const T*>(a.in);
    const T* __restrict__ hist = static_cast<const T*>(a.hist);
    const int P = a.P, M = a.M, Q = a.Q;

    if ((int)blockIdx.x == a.nblocks) {
        T* __restrict__ hn = static_cast<T*>(a.hist_next);
        for (int i = t; i < P; i += NT) {
            const long long g = a.count - P + i;
            T v;
            if (g < 0) {
                v = hist[g + P];
            } else {
                v = in[g];
                if (ROT) v = rotate(v, phasor_fx(a.phase0 + (unsigned long long)g * a.dphase), g, a.gm1);
            }
            hn[i] = v;
        }
        return;
    }

    constexpr int TJ = NT * R;                               // j values per tile (TJ*L outputs)
    const long long j0 = (long long)blockIdx.x * TJ;
    const long long base = j0 * M - P;                       // stream position of staged element 0
    const int V = TJ + Q + 1;                                // elements per branch (e_c < M adds at most one)
    const int U = V * M;
    stage_tile<CH, NT, ROT>(in, hist, P, a.count, base, U, a, [&](int u, T v) {
        const int vv = u / M, m = u - vv * M;
        lds[m * a.sb + vv] = v;
    }, (int)blockIdx.x);
    __syncthreads();

    T acc[L][R];
#pragma unroll
    for (int c = 0; c < L; c++)
#pragma unroll
        for (int p = 0; p < R; p++) acc[c][p] = Smp<CH>::zero();
    if (M == 1) {
        // pure interpolation: every sub-filter reads the SAME window (e_c = 0), so one LDS read feeds L*R
        // FMAs; taps come transposed ([q][c]: L consecutive scalars per tap position)
        const T* B = lds + t * R;
        const float* __restrict__ ht = a.taps_t;
        T win[R];
#pragma unroll
        for (int p = 0; p < R; p++) win[p] = B[p];
        for (int q0 = 0; q0 < Q; q0 += R) {
            const T* Bn = B + q0 + R;
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (q0 + j < Q) {
#pragma unroll
                    for (int c = 0; c < L; c++) {
                        const float h = ht[(q0 + j) * L + c];
#pragma unroll
                        for (int p = 0; p < R; p++) mac(acc[c][p], h, win[(p + j) % R]);
                    }
                    win[j] = Bn[j];
                }
            }
        }
    } else
#pragma unroll
    for (int c = 0; c < L; c++) {
        for (int m = 0; m < M; m++) {
            // tap k = q*M + m of sub-filter c meets staged sample e_c + M*j + k: branch (e_c + m) % M, element j + q + (e_c + m) / M
            const int em = a.e[c] + m;
            const int ob = em / M, mb = em - ob * M;
            const T* B = lds + mb * a.sb + t * R + ob;
            const float* __restrict__ hp = a.taps + ((size_t)c * M + m) * Q;
            T win[R];
#pragma unroll
            for (int p = 0; p < R; p++) win[p] = B[p];
            int q0 = 0;
            for (; q0 + R <= Q; q0 += R) {
                const T* Bn = B + q0 + R;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const float h = hp[q0 + j];
#pragma unroll
                    for (int p = 0; p < R; p++) mac(acc[c][p], h, win[(p + j) % R]);
                    win[j] = Bn[j];
                }
            }
            if (q0 < Q) {
                const T* Bn = B + q0 + R;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    if (q0 + j < Q) {
                        const float h = hp[q0 + j];
#pragma unroll
                        for (int p = 0; p < R; p++) mac(acc[c][p], h, win[(p + j) % R]);
                        win[j] = Bn[j];
                    }
                }
            }
        }
    }

    // The lane holds R*L consecutive outputs; stored straight from registers every instruction would
    // write 64 separate 8-byte pieces R*L*8 bytes apart (L2-request-bound, as the channelizer's first
    // version was).  Pass them through LDS (the staged input is dead) and store coalesced.
    __syncthreads();
#pragma unroll
    for (int p = 0; p < R; p++)
#pragma unroll
        for (int c = 0; c < L; c++) lds[(t * R + p) * L + c] = acc[c][p];
    __syncthreads();
    T* __restrict__ out = static_cast<T*>(a.out);
    const long long n0 = j0 * L;                               // first output of the tile
    long long rem = a.nout - n0;
    if (rem > (long long)TJ * L) rem = (long long)TJ * L;
    for (int i = t; i < (int)rem; i += NT) out[n0 + i] = lds[i];
}

// ---- stand-alone NCO mixer ------------------------------------------------------------
struct XlateArgs {
    const float2* in;
    float2* out;
    long long count;
    unsigned long long phase0, dphase;
    double2 rot_one;     // exp(j*2pi*dphase)
    double2 rot_stride;  // exp(j*2pi*stride*dphase), stride = 2*gridDim.x*NT samples
    int vec;             // 1: in/out 16-byte aligned -> float4 (two samples) per lane
    float gm1;           // |phase_inc| - 1, 0 = ideal NCO
};

// FrequencyXlator<complex_t>::run (src/dsp/processing.h:64): y[n] = x[n] * phase_n.
template <int NT> __global__ __launch_bounds__(NT) void xlate_kernel(const XlateArgs a) {
    const long long npairs = (a.count + 1) >> 1;
    const long long stride = (long long)gridDim.x * NT;
    long long p = (long long)blockIdx.x * NT + threadIdx.x;
    if (p >= npairs) return;
    double2 ph = phasor_fx(a.phase0 + (unsigned long long)(2 * p) * a.dphase);
    for (; p < npairs; p += stride) {
        const long long g = 2 * p;
        const double2 ph1 = cmul(ph, a.rot_one);
        if (g + 1 < a.count) {
            float2 x0, x1;
            if (!a.in) {   // SineSource: the rotator runs over a buffer of ones (src/dsp/source.h:55-59)
                x0 = make_float2(1.0f, 0.0f);
                x1 = x0;
            } else if (a.vec) {
                const float4 v = reinterpret_cast<const float4*>(a.in)[p];
                x0 = make_float2(v.x, v.y);
                x1 = make_float2(v.z, v.w);
            } else {
                x0 = a.in[g];
                x1 = a.in[g + 1];
            }
            const float2 y0 = rotate(x0, ph, g, a.gm1), y1 = rotate(x1, ph1, g + 1, a.gm1);
            if (a.vec) {
                reinterpret_cast<float4*>(a.out)[p] = make_float4(y0.x, y0.y, y1.x, y1.y);
            } else {
                a.out[g] = y0;
                a.out[g + 1] = y1;
            }
        } else {
            a.out[g] = rotate(a.in ? a.in[g] : make_float2(1.0f, 0.0f), ph, g, a.gm1);
        }
        ph = cmul(ph, a.rot_stride);
    }
}

// ---- synthetic IQ ---------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ float synth_value(uint64_t c, uint32_t key) {
    uint32_t h = mix32((uint32_t)c ^ key);
    h = mix32(h + (uint32_t)(c >> 32) * 0x9e3779b9U);
    return (float)((int32_t)(h >> 8) - (1 << 23)) * (1.0f / (float)(1 << 23));
}

// Same counter-based generator as oracle_synth_iq(); one float4 (two samples) per lane.
template <int NT>
__global__ __launch_bounds__(NT) void synth_iq_kernel(float* out, long long first_sample,
                                                      long long count, uint32_t key) {
    const long long nf = 2 * count;
    const long long stride = (long long)gridDim.x * NT * 4;
    for (long long i = ((long long)blockIdx.x * NT + threadIdx.x) * 4; i < nf; i += stride) {
        const uint64_t c = (uint64_t)(2 * first_sample + i);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = synth_value(c + j, key);
        if (i + 4 <= nf && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int j = 0; j < 4 && i + j < nf; j++) out[i + j] = v[j];
        }
    }
}

}  // namespace qk
