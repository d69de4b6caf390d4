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
//            -> channels c0 + 16 c1 at time n': rot_c(n') and stores; each store instruction writes 4
//            runs of 16 consecutive n' (128-byte lines).  (The first version scattered 64 8-byte pieces
//            per store and was bound by L2 write requests: 134 M per GiB of output.)
// rot_c(n') = A_c(tile) * W(n' - n0) * B_c(n' - n0): A_c is FP64 state of lane c advanced by one
// complex multiply per tile, W = exp(j 64 m dphi_0) a per-lane constant, B_c = exp(j 64 m delta_c) a
// 3-term FP32 series.  History is raw input.
#include "chan.hip.h"
#include "cfft.hip.h"

namespace qk {

constexpr int kChK = 64;                 // channels == branches == decimation
constexpr int kChT = 16;                 // output times per wave tile
constexpr int kChRowT = 68;              // T[n'][mu] row pitch: lanes (n', sub) read 4n' + sub (mod 32): conflict-free
constexpr int kChWaveLds = kChT * kChRowT + 2 * 64;   // float2 elements per wave: T + {A_c, theta_c, gm1_c}[64]
constexpr int kChRows = kChT - 1 + 4;    // 19 staged rows at 256 taps

template <int CTRL> __device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <bool INV>   // INV: channel c sits at +c/64 turn per sample relative to channel 0, else -c/64
__global__ __launch_bounds__(256, 3) void chan_uniform_kernel(const ChanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const int P = a.P, Q = a.Q;
    const float2* __restrict__ in = a.in;

    if ((int)blockIdx.x == a.nwg) {
        // history for the next call: the last P samples of hist ++ in
        for (int i = t; i < P; i += 256) {
            const long long g = a.count - P + i;
            a.hist_next[i] = g < 0 ? a.hist[g + P] : in[g];
        }
        return;
    }

    const int l = t & 63, wv = t >> 6;
    float2* T = lds + wv * kChWaveLds;                                   // [16][68]
    float4* tab = reinterpret_cast<float4*>(T + kChT * kChRowT);         // [64] {A_c.re, A_c.im, theta_c, gm1_c}

    // consecutive tiles on the same XCD (workgroups are dealt round-robin over the 8 XCDs): neighbours
    // share 3 of their 19 rows
    const int per_xcd = a.nwg >> 3;
    const int wg = (a.nwg & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    const int gw = wg * 4 + wv, nwaves = a.nwg * 4;
    const long long tile_pos = (long long)kChT * kChK;                    // stream positions per tile

    // ---- branch role: lane = staged column p ------------------------------------------------------
    float2 g[4];
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = q < Q ? a.gtaps[64 * q + l] : make_float2(0.0f, 0.0f);
    const int mu = (l - P) & 63;
    // ---- DFT role: lane = (n' = l >> 2, sub = l & 3) ------------------------------------------------
    const int nq = l >> 2, sub = l & 3;
    float2* twl = lds + 4 * kChWaveLds;              // [sub][c0] = exp(+-j 2pi c0 sub / 64), shared by the 4 waves
    if (t < 64) {
        float2 w = a.tw64[((t & 15) * (t >> 4)) & 63];   // exp(-j 2pi m / 64); conjugated when INV
        if (INV) w.y = -w.y;
        twl[t] = w;
    }
    __syncthreads();                                 // the only workgroup barrier: once per launch
    const float2* __restrict__ tw = twl + sub * 16;
    const int c1 = ((sub & 1) << 1) | (sub >> 1);    // radix-4 output this lane keeps (bit-reversed quad index)
    const float sA = (sub & 2) ? -1.0f : 1.0f, sB = (sub & 1) ? -1.0f : 1.0f;
    const bool is3 = sub == 3;
    float2 W;
    {
        const double2 w = fx_phasor((unsigned long long)(kChK * nq) * a.dphase0);
        W = make_float2((float)w.x, (float)w.y);
    }
    // ---- channel role: lane = channel c ----------------------------------------------------------
    double2 corr, corr_step;
    float theta_c, gm1_c;
    {
        const long long del = a.ddelta[l];                      // delta_c (tiny, signed)
        const unsigned long long inc = a.dphase0 + (unsigned long long)del;
        const long long j0 = (long long)gw * tile_pos - P;
        corr = fx_phasor(a.phase0 + a.dphi[l] + (unsigned long long)j0 * a.dphase0 + (unsigned long long)((j0 + a.kcentre) * del));
        corr_step = fx_phasor((unsigned long long)(tile_pos * nwaves) * inc);
        theta_c = (float)((double)(del * (long long)kChK) * 3.4061215800865545e-19);   // 2pi / 2^64
        gm1_c = a.gm1[l];
    }

    const float2* __restrict__ in_or_hist = a.count > 0 ? in : a.hist;   // any readable address (P >= 1)
    for (int wt = gw; wt < a.ntiles; wt += nwaves) {
        const long long n0 = (long long)wt * kChT;            // first output time of the tile
        const long long jb = n0 * kChK - P;                    // stream position of row 0, column 0
        // ---- rows from global memory ------------------------------------------------------------
        float2 x[kChRows];
        if (jb >= 0 && jb + 64 * kChRows <= a.count) {
            // interior tile: 19 independent coalesced loads (rows beyond 15 + Q meet zero taps)
            const float2* __restrict__ src = in + jb + l;
#pragma unroll
            for (int r = 0; r < kChRows; r++) x[r] = src[64 * r];
        } else {
            // first / last tiles: history in front, zeros behind; still branch-free per row
#pragma unroll
            for (int r = 0; r < kChRows; r++) {
                const long long gpos = jb + 64 * r + l;
                const bool ok = gpos >= -(long long)P && gpos < a.count;
                const float2* __restrict__ src = gpos < 0 ? a.hist + (gpos + P) : in + gpos;
                const float2 v = *(ok ? src : in_or_hist);
                x[r] = ok ? v : make_float2(0.0f, 0.0f);
            }
        }
        tab[l] = make_float4((float)corr.x, (float)corr.y, theta_c, gm1_c);
        corr = dcmul(corr, corr_step);
        // ---- branch sums: U[n'][mu] = sum_q g[64q + p] x[row n'+q][p] -------------------------------
#pragma unroll
        for (int n = 0; n < kChT; n++) {
            float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < Q) {
                    acc.x = fmaf(g[q].x, x[n + q].x, acc.x);
                    acc.x = fmaf(-g[q].y, x[n + q].y, acc.x);
                    acc.y = fmaf(g[q].x, x[n + q].y, acc.y);
                    acc.y = fmaf(g[q].y, x[n + q].x, acc.y);
                }
            }
            T[n * kChRowT + mu] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 64-point DFT over mu = sub + 4i: radix-16 in registers, radix-4 across the quad ----------
        float2 U[16];
#pragma unroll
        for (int i = 0; i < 16; i++) U[i] = T[nq * kChRowT + sub + 4 * i];
        fft16<INV>(U);                               // over i -> group c0 at U[rev16(c0)]
        const long long nn = n0 + nq;
        const long long j = nn * kChK - P + a.kcentre;               // window-centre position of this output
        const float jm = (float)(int)(j & 511);
        const float fl = (float)nq;
        float2* __restrict__ o = a.out + nn;
        const bool live = nn < a.nout;
#pragma unroll
        for (int c0 = 0; c0 < 16; c0++) {
            float2 z = U[rev16(c0)];
            if (c0 != 0) z = cmulc<false>(z, tw[c0]);
            // stage A: pairs (sub, sub^2); lane 3 takes the +-j twiddle
            float2 ta = make_float2(fmaf(sA, z.x, dpp_quad<0x4E>(z.x)), fmaf(sA, z.y, dpp_quad<0x4E>(z.y)));
            {
                const float2 tj = mulj<INV>(ta);
                ta.x = is3 ? tj.x : ta.x;
                ta.y = is3 ? tj.y : ta.y;
            }
            // stage B: pairs (sub, sub^1)
            const float2 y = make_float2(fmaf(sB, ta.x, dpp_quad<0xB1>(ta.x)), fmaf(sB, ta.y, dpp_quad<0xB1>(ta.y)));
            // rot_c(n') and VOLK's magnitude sawtooth
            const int c = c0 + 16 * c1;
            const float4 tc = tab[c];
            const float ang = fl * tc.z, a2 = ang * ang;
            const float br = fmaf(a2, -0.5f, 1.0f), bi = fmaf(a2 * ang, -1.0f / 6.0f, ang);
            const float gain = fmaf(jm, tc.w, 1.0f);
            const float2 aw = cmulc<false>(make_float2(tc.x, tc.y), W);
            const float pr = fmaf(aw.x, br, -aw.y * bi) * gain, pi = fmaf(aw.x, bi, aw.y * br) * gain;
            const float2 r = make_float2(fmaf(y.x, pr, -y.y * pi), fmaf(y.x, pi, y.y * pr));
            if (live) o[(size_t)c * a.out_stride] = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

size_t chan_uniform_lds_bytes() { return (size_t)(4 * kChWaveLds + 64) * sizeof(float2); }

int launch_chan_uniform(const ChanArgs& a, int grid, hipStream_t stream) {
    const size_t lds_bytes = chan_uniform_lds_bytes();
    if (a.inv) hipLaunchKernelGGL(chan_uniform_kernel<true>, dim3(grid), dim3(256), lds_bytes, stream, a);
    else hipLaunchKernelGGL(chan_uniform_kernel<false>, dim3(grid), dim3(256), lds_bytes, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
