// mf_dec.hip -- MFMA decimator for large integer decimations (design notes: mf_dec.hip.h).
#include "mf_dec.hip.h"
#include <type_traits>
#include "kernels.hip.h"

namespace qk {

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;

// value of `src` in the lane the DPP control pairs this one with; lanes whose source is outside the row read 0, lanes
// in rows outside ROWMASK keep `old`
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROWMASK, 0xf, true));
}
using f32x2 = __attribute__((ext_vector_type(2))) float;
// complex product on the packed FP32 pipe: two instructions (the operand swizzle and the sign ride on op_sel / neg_lo;
// written out in C the compiler spends six on it).  _s: the second factor is wave-uniform (SGPR pair).
__device__ __forceinline__ f32x2 cmul_pk(f32x2 x, f32x2 p) {
    f32x2 t, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(p));                                    // (xr pr, xr pi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(o) : "v"(x), "v"(p), "v"(t));   // (-xi pi, xi pr) + t
    return o;
}
__device__ __forceinline__ f32x2 cmul_pk_s(f32x2 x, f32x2 p) {
    f32x2 t, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "s"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(o) : "v"(x), "s"(p), "v"(t));
    return o;
}
constexpr int kShl = 0x100, kShr = 0x110;      // row_shl:n -- lane i reads lane i + n;  row_shr:n -- lane i reads lane i - n

// Diagonal sums of a 16 x 16 result tile.  d[v] = Z[4g + v][rho] on lane 16 g + rho.  cur[rho] collects the terms
// Z[q][rho + q] with rho + q < 16 (output rho of THIS tile), prev[rho] the terms with rho + q >= 16, i.e. Z[q][rho + q - 16]
// (output rho of the PREVIOUS tile); both still spread over the four lane groups.
__device__ __forceinline__ void diag_sum(const f32x4 d, float& cur, float& prev) {
    // within a group: shift by the register index
    float i_ = d[0] + dpp<kShl + 1>(0.0f, d[1]);
    i_ += dpp<kShl + 2>(0.0f, d[2]);
    i_ += dpp<kShl + 3>(0.0f, d[3]);                           // I[e] = sum_v Z[4g+v][e + v], e + v < 16
    float r_ = dpp<kShr + 15>(0.0f, d[1]);
    r_ += dpp<kShr + 14>(0.0f, d[2]);
    r_ += dpp<kShr + 13>(0.0f, d[3]);                          // R[e + 16] = the same for e = -3..-1
    // per group: shift by 4 g
    float c = i_;
    c = dpp<kShl + 4, 0x2>(c, i_);
    c = dpp<kShl + 8, 0x4>(c, i_);
    c = dpp<kShl + 12, 0x8>(c, i_);
    float p = r_;
    p = dpp<kShl + 4, 0x2>(p, r_);
    p = dpp<kShl + 8, 0x4>(p, r_);
    p = dpp<kShl + 12, 0x8>(p, r_);
    float w = dpp<kShr + 12, 0x2>(0.0f, i_);
    w = dpp<kShr + 8, 0x4>(w, i_);
    w = dpp<kShr + 4, 0x8>(w, i_);
    cur = c;
    prev = p + w;
}
}  // namespace

// rot_k: the NCO table exp(j 64 i dphase) -- the kernel arguments' (one channel) or the channel's row of the device table
// RD (round 4): REAL data -- PolyphaseResampler<float> with interp 1 (src/dsp/resampling.h:113-118).  A sample is one float: the same rows, the
// same A operands (tapk) and k order, half the bytes, half the LDS traffic and only the "re" products -- the even and the odd columns of a step
// accumulate in two registers sets (two independent MFMA chains, as re / im are in the complex form) that are added at the end.
template <int KJ, bool ROT, int DEPTH, int QS, bool RD = false> __device__ __forceinline__ void decim_mfma_body(const MfArgs& a, const float2* __restrict__ rot_k) {
    static_assert(!(RD && ROT), "the NCO is a complex operator");
    using S = typename std::conditional<RD, float, float2>::type;      // one sample
    // row pitch in samples: K + 2 complex = 4 x odd dwords (every ds_read_b128 group on 16 distinct 4-bank groups); K + 4 real = 4 x odd dwords
    // as well (the ds_read_b64 of 32 lanes -- 16 rows x 2 column pairs -- on 32 distinct bank pairs)
    constexpr int K = 8 * KJ, PITCH = RD ? K + 4 : K + 2, NI = 2 * KJ;
    const int t = threadIdx.x, l = t & 63;
    const int P = a.P, M = a.M;
    const S* __restrict__ in_s = reinterpret_cast<const S*>(a.in);
    const S* __restrict__ hist_s = reinterpret_cast<const S*>(a.hist);
    S* __restrict__ out_s = reinterpret_cast<S*>(a.out);
    auto zero_s = [] { if constexpr (RD) return 0.0f; else return make_float2(0.0f, 0.0f); };
    if ((int)blockIdx.x == (a.ntasks + 3) / 4) {
        // history hand-over (resampling.h:129): last P samples of hist ++ in, rotated for the fused VFO
        S* hist_next_s = reinterpret_cast<S*>(a.hist_next);
        for (int i = t; i < P; i += 256) {
            const long long g = a.count - P + i;
            S v;
            if (g < 0) v = hist_s[g + P];
            else {
                v = in_s[g];
                if constexpr (ROT) v = rotate(v, phasor_fx(a.phase0 + (unsigned long long)g * a.dphase), g, a.gm1);
            }
            hist_next_s[i] = v;
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) S tile_all[4][16 * PITCH + 64];      // + a spare slot per lane
    S* tile = tile_all[t >> 6];
    const int task = (int)blockIdx.x * 4 + (t >> 6);
    if (task >= a.ntasks) return;
    const long long n0 = (long long)task * a.T;
    long long n1 = n0 + a.T;
    if (n1 > a.nout) n1 = a.nout;
    const int ntiles = (int)((n1 - n0 + 15) >> 4) + QS;       // + QS: the rows the last outputs reach into (16 per tap set)
    const int E = 16 * M;                                     // samples per tile

    // columns M..K-1 of the rows are never written: they meet zero taps, but must hold finite values
    for (int i = l; i < 16 * (K - M); i += 64) {
        const int row = i / (K - M);
        tile[row * PITCH + M + (i - row * (K - M))] = zero_s();
    }

    // tap set s = tap rows 16 s .. 16 s + 15 (QS = 2: up to 32 taps per column)
    float ta[QS][KJ][2];
#pragma unroll
    for (int s = 0; s < QS; s++)
#pragma unroll
        for (int jj = 0; jj < KJ; jj++) {
            ta[s][jj][0] = a.tapk[((s * KJ + jj) * 2) * 64 + l];
            ta[s][jj][1] = a.tapk[((s * KJ + jj) * 2 + 1) * 64 + l];
        }
    // element i of a tile for this lane: sample 64 i + l of its 16 M, LDS slot [row][col]; the elements past the tile's end
    // (only the last two loads can reach there) re-read its last sample and park it in the lane's spare slot
    int woff[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int idx = 64 * i + l;
        const int row = (int)__umulhi((unsigned)idx, a.minv);          // idx / M (exact: idx < 2^11, M <= 128)
        woff[i] = idx < E ? row * PITCH + (idx - row * M) : 16 * PITCH + l;
    }
    const int e1 = min(64 * (NI - 2) + l, E - 1), e2 = min(64 * (NI - 1) + l, E - 1);
    // B operand of this lane: row l % 16, columns 8 jj + 2 (l / 16) + {0, 1} -- one ds_read_b128 (complex) / ds_read_b64 (real) per step
    using B2 = typename std::conditional<RD, float2, float4>::type;
    const B2* brd = reinterpret_cast<const B2*>(tile + (l & 15) * PITCH + 2 * (l >> 4));

    const long long gbase = (long long)M * n0 - P;            // sample index (relative to in[0]) of tile 0's element 0
    double2 pd;
    if (ROT) pd = phasor_fx(a.phase0 + (unsigned long long)(gbase + l) * a.dphase);

    // DEPTH tiles are in flight in registers ahead of the one being multiplied
    S xb[DEPTH][NI];
    bool plainb[DEPTH];
    const bool tail = 64 * (NI - 1) < E;                      // wave-uniform: the last load holds samples of the tile at all
    auto tile_plain = [&](int tt) {                           // wave-uniform: the whole tile is plain input
        const long long g0 = gbase + (long long)E * tt;
        return tt < ntiles && g0 >= 0 && g0 + E <= a.count;
    };
    auto load_tile = [&](int tt, S (&xn)[NI]) {
        const S* __restrict__ p = in_s + (gbase + (long long)E * tt);
#pragma unroll
        for (int i = 0; i < NI - 2; i++) xn[i] = p[64 * i + l];
        xn[NI - 2] = p[e1];
        if (tail) xn[NI - 1] = p[e2];
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        plainb[d] = tile_plain(d);
        if (plainb[d]) load_tile(d, xb[d]);
    }
    float carry_re = 0.0f, carry_im = 0.0f, carry2_re = 0.0f, carry2_im = 0.0f;
    // The outputs of a tile are stored one tile LATER, behind an explicit wait for the next tile's samples.  gfx950 counts loads and stores on
    // one counter (vmcnt) and the store sits behind a lane test, so hipcc's wait for the next tile's samples is vmcnt(0): issued at the end of
    // its own tile the store was the youngest request in flight at that wait and every tile paid a write round trip (rounds 2-3).  Issued
    // here it is a whole tile's arithmetic old when the next wait comes (profiles/r04_mf_deferred_store.txt: -1 ... -10 % from 2^22 samples on; a
    // reference-sized call, two or three tiles per wave, is 0.2 us quicker the old way: a.defer_store, wave-uniform).
    float pend_re = 0.0f, pend_im = 0.0f;
    auto store_tile = [&](int tt) {                           // the outputs tile tt completed (those of tile tt - QS)
        const long long n = n0 + 16LL * (tt - QS) + l;
        if (tt >= QS && l < 16 && n < n1) {
            // keep2: the kernel runs at half the decimation (rows of M / 2 samples: decimations 130-256) and every other output is the call's
            S o;
            if constexpr (RD) o = pend_re; else o = make_float2(pend_re, pend_im);
            if (!a.keep2) out_s[n] = o;
            else if (!(n & 1)) out_s[n >> 1] = o;
        }
    };
    auto do_tile = [&](int tt, S (&xn)[NI], bool& plain) {
        const long long g0 = gbase + (long long)E * tt;
        if (a.defer_store) {
            __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): this tile's samples (and stores that are a tile old)
            if (tt > 0) store_tile(tt - 1);
        }
        // NCO: lane phasor of the tile (FP64 recurrence, rounded once) x the FP32 table exp(j 64 i dphase) inside it,
        // x VOLK's magnitude sawtooth 1 + (g mod 512) gm1 (rotate(), kernels.hip.h): g advances by 64 per load, so the
        // sawtooth takes 8 values per lane and tile, folded into 8 copies of the lane phasor
        f32x2 pg[8];
        if constexpr (ROT) {
            const f32x2 pf = {(float)pd.x, (float)pd.y};
            const int m0 = (int)((g0 + l) & 511);
#pragma unroll
            for (int k = 0; k < 8; k++) pg[k] = pf * fmaf((float)((m0 + 64 * k) & 511), a.gm1, 1.0f);
        }
        auto spin = [&](float2 v, int i) {
            const f32x2 w = {rot_k[i].x, rot_k[i].y};
            const f32x2 o = cmul_pk(f32x2{v.x, v.y}, cmul_pk_s(pg[i & 7], w));
            return make_float2(o.x, o.y);
        };
        if (plain) {
#pragma unroll
            for (int i = 0; i < NI; i++)
                if (i < NI - 1 || tail) {
                    if constexpr (ROT) tile[woff[i]] = spin(xn[i], i);
                    else tile[woff[i]] = xn[i];
                }
        } else {
            // a tile that touches the history or the end of the call (the first and the last of a call): rolled, guarded
#pragma unroll 1
            for (int i = 0; i < NI; i++) {
                const int idx = 64 * i + l;
                if (idx < E) {
                    const long long g = g0 + idx;
                    S v = zero_s();
                    if (g < 0) { if (g + P >= 0) v = hist_s[g + P]; }               // (history is already rotated)
                    else if (g < a.count) {
                        v = in_s[g];
                        if constexpr (ROT) v = rotate_f(v, make_float2((float)pd.x, (float)pd.y), rot_k[i], g, a.gm1);
                    }
                    const int row = (int)__umulhi((unsigned)idx, a.minv);
                    tile[row * PITCH + (idx - row * M)] = v;
                }
            }
        }
        if (ROT) pd = cmul(pd, a.rot_step);
        plain = tile_plain(tt + DEPTH);
        if (plain) load_tile(tt + DEPTH, xn);

        f32x4 zr[QS], zi[QS];
#pragma unroll
        for (int s = 0; s < QS; s++) {
            zr[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            zi[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int jj = 0; jj < KJ; jj++) {
            const B2 b = brd[4 * jj];                                            // 8 samples further per step
#pragma unroll
            for (int s = 0; s < QS; s++) {
                if constexpr (RD) {
                    zr[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][0], b.x, zr[s], 0, 0, 0);      // even columns
                    zi[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][1], b.y, zi[s], 0, 0, 0);      // odd columns: the second chain
                } else {
                    zr[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][0], b.x, zr[s], 0, 0, 0);
                    zi[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][0], b.y, zi[s], 0, 0, 0);
                    zr[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][1], b.z, zr[s], 0, 0, 0);
                    zi[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][jj][1], b.w, zi[s], 0, 0, 0);
                }
            }
        }
        if constexpr (RD) {
#pragma unroll
            for (int s = 0; s < QS; s++) zr[s] += zi[s];
        }
        float cur_re, prev_re, cur_im = 0.0f, prev_im = 0.0f;
        diag_sum(zr[0], cur_re, prev_re);
        if constexpr (!RD) diag_sum(zi[0], cur_im, prev_im);
        float o_re, o_im;
        if (QS == 1) {
            o_re = carry_re + prev_re;                                           // outputs of the previous tile, complete
            o_im = carry_im + prev_im;
            carry_re = cur_re;
            carry_im = cur_im;
        } else {
            // the second tap set reaches 16 rows further back: its in-tile part belongs to the previous tile's outputs, its
            // carried part to the tile before that, which this tile completes
            float c1_re, p1_re, c1_im = 0.0f, p1_im = 0.0f;
            diag_sum(zr[QS - 1], c1_re, p1_re);
            if constexpr (!RD) diag_sum(zi[QS - 1], c1_im, p1_im);
            o_re = carry2_re + p1_re;
            o_im = carry2_im + p1_im;
            carry2_re = carry_re + prev_re + c1_re;
            carry2_im = carry_im + prev_im + c1_im;
            carry_re = cur_re;
            carry_im = cur_im;
        }
        o_re += __shfl_xor(o_re, 16);
        if constexpr (!RD) o_im += __shfl_xor(o_im, 16);
        o_re += __shfl_xor(o_re, 32);
        if constexpr (!RD) o_im += __shfl_xor(o_im, 32);
        pend_re = o_re;
        pend_im = o_im;
        if (!a.defer_store) store_tile(tt);
    };
    for (int tt = 0; tt < ntiles; tt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (tt + d < ntiles) do_tile(tt + d, xb[d], plainb[d]);
    }
    if (a.defer_store && ntiles > 0) store_tile(ntiles - 1);
}

template <int KJ, bool ROT, int DEPTH, int QS>
__global__ __launch_bounds__(256, KJ * DEPTH * QS <= 8 ? 4 : (KJ * DEPTH * QS <= 12 ? 3 : 2)) void decim_mfma_kernel(const MfArgs a) {
    decim_mfma_body<KJ, ROT, DEPTH, QS>(a, a.rot_k);
}
template <int KJ, int DEPTH, int QS>
__global__ __launch_bounds__(256, KJ * DEPTH * QS <= 8 ? 4 : (KJ * DEPTH * QS <= 14 ? 3 : 2)) void decim_mfma_real_kernel(const MfArgs a) {
    decim_mfma_body<KJ, false, DEPTH, QS, true>(a, a.rot_k);
}

template <int KJ, int DEPTH>
__global__ __launch_bounds__(256, KJ * DEPTH <= 8 ? 4 : (KJ * DEPTH <= 12 ? 3 : 2)) void decim_mfma_batch_kernel(const MfBatchArgs b) {
    const int ch = blockIdx.y;
    MfArgs a = b.a;
    const MfChanConst& c = b.tab[ch];
    a.hist = c.hist[b.cur];
    a.hist_next = const_cast<float2*>(c.hist[b.cur ^ 1]);
    a.out = b.use_ptrs ? static_cast<float2*>(b.outs[ch]) : b.a.out + (long long)ch * b.out_stride;
    a.phase0 = b.phase0[ch];
    a.dphase = c.dphase;
    a.rot_step = c.rot_step;
    a.gm1 = c.gm1;
    decim_mfma_body<KJ, true, DEPTH, 1>(a, c.rot_k);
}

// Two tiles in flight (DEPTH 2) is instantiated up to KJ = 12 only: at KJ = 16 the 2 x 16 prefetch registers push the kernel to
// 256 VGPRs plus scratch (round 2: 28-32 bytes per lane).  AUTO asks for DEPTH 2 up to KJ = 8 (qdsp_hip.hip); QDSP_HIP_MF_DEPTH = 2 on
// longer rows gets DEPTH 1 beyond 12.
constexpr int kMfDepth2MaxKJ = 12;

template <int K> static int launch_mf_batch_k(const MfBatchArgs& b, dim3 grid, int depth, hipStream_t stream) {
    const dim3 block(256);
    if constexpr (K <= kMfDepth2MaxKJ) {
        if (depth == 2) { hipLaunchKernelGGL((decim_mfma_batch_kernel<K, 2>), grid, block, 0, stream, b); return 0; }
    }
    hipLaunchKernelGGL((decim_mfma_batch_kernel<K, 1>), grid, block, 0, stream, b);
    return 0;
}

int launch_mf_dec_batch(const MfBatchArgs& b, int nchan, int KJ, int depth, hipStream_t stream) {
    const dim3 grid((b.a.ntasks + 3) / 4 + 1, nchan);
#define QK_MFB(k)                                           \
    if (KJ == k) {                                          \
        launch_mf_batch_k<k>(b, grid, depth, stream);       \
        const hipError_t e = hipGetLastError();             \
        return e == hipSuccess ? 0 : -(int)e;               \
    }
    QK_MFB(2) QK_MFB(3) QK_MFB(4) QK_MFB(5) QK_MFB(6) QK_MFB(7) QK_MFB(8)
    QK_MFB(9) QK_MFB(10) QK_MFB(11) QK_MFB(12) QK_MFB(13) QK_MFB(14) QK_MFB(15) QK_MFB(16)
#undef QK_MFB
    return -1;
}

template <int K> static void launch_mf_k(const MfArgs& a, dim3 grid, bool rot, int depth, int qs, bool real, hipStream_t stream) {
    const dim3 block(256);
    if (real) {
        if (qs == 2) { hipLaunchKernelGGL((decim_mfma_real_kernel<K, 1, 2>), grid, block, 0, stream, a); return; }
        if constexpr (K <= kMfDepth2MaxKJ) {
            if (depth == 2) { hipLaunchKernelGGL((decim_mfma_real_kernel<K, 2, 1>), grid, block, 0, stream, a); return; }
        }
        hipLaunchKernelGGL((decim_mfma_real_kernel<K, 1, 1>), grid, block, 0, stream, a);
        return;
    }
    if (qs == 2 && rot) { hipLaunchKernelGGL((decim_mfma_kernel<K, true, 1, 2>), grid, block, 0, stream, a); return; }
    if (qs == 2) { hipLaunchKernelGGL((decim_mfma_kernel<K, false, 1, 2>), grid, block, 0, stream, a); return; }
    if constexpr (K <= kMfDepth2MaxKJ) {
        if (depth == 2 && rot) { hipLaunchKernelGGL((decim_mfma_kernel<K, true, 2, 1>), grid, block, 0, stream, a); return; }
        if (depth == 2) { hipLaunchKernelGGL((decim_mfma_kernel<K, false, 2, 1>), grid, block, 0, stream, a); return; }
    }
    if (rot) hipLaunchKernelGGL((decim_mfma_kernel<K, true, 1, 1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((decim_mfma_kernel<K, false, 1, 1>), grid, block, 0, stream, a);
}

int launch_mf_dec(const MfArgs& a, int KJ, bool rot, int depth, int qs, bool real, hipStream_t stream) {
    if (real && rot) return -1;
    const dim3 grid((a.ntasks + 3) / 4 + 1);
#define QK_MF(k)                                            \
    if (KJ == k) {                                          \
        launch_mf_k<k>(a, grid, rot, depth, qs, real, stream); \
        const hipError_t e = hipGetLastError();             \
        return e == hipSuccess ? 0 : -(int)e;               \
    }
    QK_MF(2) QK_MF(3) QK_MF(4) QK_MF(5) QK_MF(6) QK_MF(7) QK_MF(8)
    QK_MF(9) QK_MF(10) QK_MF(11) QK_MF(12) QK_MF(13) QK_MF(14) QK_MF(15) QK_MF(16)
#undef QK_MF
    return -1;
}

}  // namespace qk
