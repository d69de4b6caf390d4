// rm_resamp.hip -- rational polyphase resampler on the MFMA units (design notes: rm_resamp.hip.h).
#include "rm_resamp.hip.h"
#include <type_traits>
#include "kernels.hip.h"

namespace qk {

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
}

// (Round 3 also tried two tiles in flight per wave instead of one -- 24 more VGPRs, still three waves per SIMD: 147/160 0.4466 -> 0.4493 ms per
// 2^27 samples, the other ratios 2-4 % slower -- and four waves per SIMD (128 VGPRs: 2-13 spilled).  Bytes in flight are not what it waits for.)
// RD (round 4): REAL data -- PolyphaseResampler<float> (src/dsp/resampling.h:113-118).  A sample is one float: the same tiles, rows, A operands and
// band walk; the B operand of a step is one ds_read_b32, only the "re" product remains -- the even and the odd columns of a chunk accumulate in two
// register sets (two independent MFMA chains, as re / im are in the complex form), added at the end; a lane stores its four outputs as 16 bytes.
template <bool ROT, bool RD> __device__ __forceinline__ void resamp_mfma_body(const RmArgs& a) {
    static_assert(!(RD && ROT), "the NCO is a complex operator");
    using S = typename std::conditional<RD, float, float2>::type;      // one sample
    constexpr int NE = kRmNE, GM = kRmMaxGrp;
    const int t = threadIdx.x, l = t & 63;
    const int P = a.P, M = a.M, L = a.L;
    const S* __restrict__ in_s = reinterpret_cast<const S*>(a.in);
    const S* __restrict__ hist_s = reinterpret_cast<const S*>(a.hist);
    S* __restrict__ out_s = reinterpret_cast<S*>(a.out);
    auto zero_s = [] { if constexpr (RD) return 0.0f; else return make_float2(0.0f, 0.0f); };
    if ((int)blockIdx.x == (a.nwaves + 3) / 4) {
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
    extern __shared__ __attribute__((aligned(16))) unsigned char rm_smem[];
    const int rows = 4 * a.G;                                   // periods per tile
    // A operands for the whole workgroup: atab[g][k][lane] = W[4 (16 g + lane / 4) + lane % 4][cb + k], cb = first band
    // column of the lane's block (in registers they cost 72 VGPRs and the third wave per SIMD)
    float* atab = reinterpret_cast<float*>(rm_smem);
    const int na = a.ngrp * a.KB * 64, nmeta = 2 * a.ngrp * 64;
    // (... and behind them the two per-lane block tables -- first band column, period quad | output block -- which round 2 held in six VGPRs
    // per lane: read back from LDS where they are used the plain kernel fits 122 VGPRs = four waves per SIMD (measured: +- 1 % -- it is not
    // occupancy this kernel lacks) and the fused one drops from 142 to 130, which is worth 4-8 % to it: 147/160 0.240 -> 0.225 ms per 2^26 samples,
    // 3/8 0.178 -> 0.164)
    for (int i = t; i < na + nmeta; i += 256) atab[i] = a.atab[i];
    __syncthreads();
    const int* lmeta = reinterpret_cast<const int*>(atab + na);
    S* tile = reinterpret_cast<S*>(atab + ((na + nmeta + 3) & ~3)) + (size_t)(t >> 6) * (rows * a.pitch + 64);
    const int wave = (int)blockIdx.x * 4 + (t >> 6);
    if (wave >= a.nwaves) return;
    // sample e of a tile for this lane: tile-relative index u = 64 e + l, period row u / M, column u % M; a column below
    // ext is also the tail of the previous row; what has no slot goes to the lane's spare one.  (Recomputed per tile:
    // as register tables the slots cost 24 VGPRs and spills.)
    const int spare = rows * a.pitch + l;
    const long long tstep = (long long)rows * M;                // input samples per tile
    S xn[NE];
    auto tile_plain = [&](int T) {                              // wave-uniform: the whole tile is plain input
        const long long g0 = tstep * T - P;                     // sample index (relative to in[0]) of the tile's element 0
        return T < a.ntiles && g0 >= 0 && g0 + a.total <= a.count;
    };
    auto load_tile = [&](int T) {
        const S* __restrict__ p = in_s + (tstep * T - P);
#pragma unroll
        for (int e = 0; e < NE; e++)
            if (64 * e < a.total) xn[e] = p[min(64 * e + l, a.total - 1)];         // (past the tile: its last sample again, not stored)
    };
    auto put = [&](int u, S v) {
        const int b = M == 1 ? u : (int)__umulhi((unsigned)u, a.minv), c = u - b * M;      // (ceil(2^32 / 1) does not fit the multiplier)
        tile[(u < a.total && b < rows) ? b * a.pitch + c : spare] = v;
        tile[(u < a.total && b >= 1 && c < a.ext) ? (b - 1) * a.pitch + M + c : spare] = v;
    };
    int T = wave;
    double2 pd;
    if constexpr (ROT) pd = phasor_fx(a.phase0 + (unsigned long long)(tstep * T - P + l) * a.dphase);
    bool plain = tile_plain(T);
    if (plain) load_tile(T);
    // B operand of this lane: period l % 4 of a quad, columns from the block's band start
    const int brow = (l & 3) * a.pitch;
    for (; T < a.ntiles; T += a.nwaves) {
        const long long g0 = tstep * T - P;
        // ---- stage: registers -> (NCO) -> LDS rows ------------------------------------------------------------------
        {
            float2 pf;
            int m0;
            if constexpr (ROT) {
                pf = make_float2((float)pd.x, (float)pd.y);
                pd = cmul(pd, a.rot_step);
                m0 = (int)((g0 + l) & 511);
            }
            if (plain) {
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    if (64 * e < a.total) {                     // wave-uniform
                        S v = xn[e];
                        if constexpr (ROT) {
                            const float2 wk = a.rot_k[e];
                            const float gain = fmaf((float)((m0 + 64 * e) & 511), a.gm1, 1.0f);   // VOLK's magnitude sawtooth (rotate(), kernels.hip.h)
                            const float pr = fmaf(pf.x, wk.x, -pf.y * wk.y) * gain, pi = fmaf(pf.x, wk.y, pf.y * wk.x) * gain;
                            v = rot_apply(v, pr, pi);
                        }
                        put(64 * e + l, v);
                    }
                }
            } else {
                // a tile that touches the history or the end of the call (the first and the last of a call): rolled, guarded
#pragma unroll 1
                for (int e = 0; 64 * e < a.total; e++) {
                    const int u = 64 * e + l;
                    const long long g = g0 + u;
                    S v = zero_s();
                    if (u < a.total) {
                        if (g < 0) { if (g + P >= 0) v = hist_s[g + P]; }           // (history is already rotated)
                        else if (g < a.count) {
                            v = in_s[g];
                            if constexpr (ROT) v = rotate_f(v, pf, a.rot_k[e], g, a.gm1);
                        }
                    }
                    put(u, v);
                }
            }
        }
        plain = tile_plain(T + a.nwaves);
        if (plain) load_tile(T + a.nwaves);
        // ---- matrix product per quad of periods and group of 16 blocks ------------------------------------------------
        const long long per0 = (long long)rows * T;             // first period of the tile
        for (int qd = 0; qd < a.G; qd += a.qpb) {
#pragma unroll
            for (int g = 0; g < GM; g++) {
                if (g < a.ngrp) {
                    const int cbg = lmeta[g * 64 + l], mg = lmeta[(a.ngrp + g) * 64 + l];
                    const int lq = qd + (mg >> 16);             // the lane's period quad
                    const S* bp = tile + 4 * lq * a.pitch + brow + cbg;
                    const float* ap = atab + g * a.KB * 64 + l;
                    f32x4 zr = {0.0f, 0.0f, 0.0f, 0.0f}, zi = {0.0f, 0.0f, 0.0f, 0.0f};
                    // Band columns in chunks of eight: a run-time loop over the chunks, the steps of a full chunk unrolled without a test
                    // (round 3: the fully unrolled form tested `k < KB` at each of its KM steps and kept the KM wave-uniform results
                    // alive -- 109 / 166 SGPRs spilled to VGPR lanes, 211 v_readlane in the loop), the last partial chunk rolled.
                    int k0 = 0;
#pragma unroll 1
                    for (; k0 + 8 <= a.KB; k0 += 8) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const S bv = bp[k0 + k];
                            if constexpr (RD) {
                                if (k & 1) zi = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[(k0 + k) * 64], bv, zi, 0, 0, 0);
                                else zr = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[(k0 + k) * 64], bv, zr, 0, 0, 0);
                            } else {
                                zr = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[(k0 + k) * 64], bv.x, zr, 0, 0, 0);
                                zi = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[(k0 + k) * 64], bv.y, zi, 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);      // reads run at most 8 steps ahead (16 VGPRs)
                    }
#pragma unroll 1
                    for (; k0 < a.KB; k0++) {
                        const S bv = bp[k0];
                        if constexpr (RD) {
                            zr = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[k0 * 64], bv, zr, 0, 0, 0);
                        } else {
                            zr = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[k0 * 64], bv.x, zr, 0, 0, 0);
                            zi = __builtin_amdgcn_mfma_f32_4x4x1f32(ap[k0 * 64], bv.y, zi, 0, 0, 0);
                        }
                    }
                    if constexpr (RD) zr += zi;
                    // lane = (block l / 4, period l % 4 of the quad): outputs o .. o + 3 of that period, 32 bytes
                    const int o = 4 * (mg & 0xffff);
                    const long long n = (per0 + 4 * lq + (l & 3)) * L + o;
                    if (o < L && n < a.nout) {
                        S* dst = out_s + n;
                        if constexpr (RD) {
                            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // (the block starts on any sample: 4-byte aligned)
                            if (o + 3 < L && n + 3 < a.nout) {
                                *reinterpret_cast<f4u*>(dst) = (f4u){zr[0], zr[1], zr[2], zr[3]};
                            } else {
#pragma unroll
                                for (int v = 0; v < 4; v++)
                                    if (o + v < L && n + v < a.nout) dst[v] = zr[v];
                            }
                        } else if (o + 3 < L && n + 3 < a.nout) {
                            // (round 3: the lane's 32 contiguous bytes as two 16-byte stores at 8-byte alignment -- odd L puts odd periods on
                            // odd samples -- measured 0.451 against 0.442 ms for these four 8-byte ones on 147/160: not kept)
                            // (... and written non-temporally 0.601 against 0.435 ms: 8-byte pieces 32 bytes apart do not combine on that path)
                            dst[0] = make_float2(zr[0], zi[0]);
                            dst[1] = make_float2(zr[1], zi[1]);
                            dst[2] = make_float2(zr[2], zi[2]);
                            dst[3] = make_float2(zr[3], zi[3]);
                        } else {
                            if constexpr (!RD) {
#pragma unroll
                                for (int v = 0; v < 4; v++)
                                    if (o + v < L && n + v < a.nout) dst[v] = make_float2(zr[v], zi[v]);
                            }
                        }
                    }
                }
            }
        }
    }
}

template <bool ROT> __global__ __launch_bounds__(256, ROT ? 3 : 4) void resamp_mfma_kernel(const RmArgs a) { resamp_mfma_body<ROT, false>(a); }
__global__ __launch_bounds__(256, 4) void resamp_mfma_real_kernel(const RmArgs a) { resamp_mfma_body<false, true>(a); }

int launch_rm_resamp(const RmArgs& a, bool rot, bool real, hipStream_t stream) {
    if (real && rot) return -1;
    const size_t lds = rm_lds_bytes(a.ngrp, a.KB, a.G, a.pitch, real);      // <= 64 KB (checked where the plan is made)
    const dim3 grid((a.nwaves + 3) / 4 + 1), block(256);
    if (real) hipLaunchKernelGGL(resamp_mfma_real_kernel, grid, block, lds, stream, a);
    else if (rot) hipLaunchKernelGGL((resamp_mfma_kernel<true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((resamp_mfma_kernel<false>), grid, block, lds, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
