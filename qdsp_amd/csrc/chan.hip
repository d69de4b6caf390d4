// chan.hip -- uniform polyphase channelizer: 64 frequency-translating decimators in one pass.
//
// Reference shape: Splitter -> 64 x VFO (src/dsp/routing.h:47-57, src/dsp/vfo.h:19-36), i.e.
// for channel c   y_c[n'] = sum_k h[k] * x[j] * exp(j*phase_c(j)),   j = n'*M - P + k,
// with phase_c(j) = phi_c + j*dphi_c advancing by that channel's own (rounded-float) phase
// increment (src/dsp/processing.h:20,64).  When the 64 increments are uniformly spaced by
// +-1/64 turn and the decimation is M = 64, the sum factors (k = 64q + p):
//     y_c[n'] = corr_c(n') * sum_mu exp(+-j 2pi c mu / 64) * U[mu],
//     U[mu]   = sum_q h[64q + p] * xr[n'*64 - P + 64q + p],      p = (mu + P) mod 64,
//     xr[j]   = x[j] * exp(j*(phi_0 + j*dphi_0))                 (channel 0's NCO),
// a 64-branch polyphase filter followed by ONE 64-point DFT across the branches for all 64
// channels: ~40 FLOP per input sample instead of 64 x 134, and the input is read once.
// corr_c(n') carries everything that is NOT exactly uniform: the channel's carried phase
// relative to channel 0 and the (~1e-8 turn/sample) deviation of its rounded increment from
// the ideal spacing, both in exact 64-bit fixed point, applied at the centre of the tap
// window (error <= 2pi * 1e-8 * ntaps/2 at the window edges, ~1e-6 after tap weighting);
// VOLK's magnitude sawtooth (kernels.hip.h rotate()) is applied the same way per channel.
//
// One workgroup = 256 lanes = 64 output times x 64 channels per tile, persistent over tiles:
//   stage xr (4288 samples, rotated while staging) -> LDS
//   lane (n', sub): 16 branch sums U[sub + 4i] (taps in registers) -> radix-16 DFT in registers
//   LDS transpose -> lane (n', g): radix-4 across sub -> 16 channel outputs
//   LDS transpose -> wave w, lane n': correction and store of channels 16w .. 16w+15, each store
//   instruction 64 consecutive n' of one channel (512 contiguous bytes: the first version stored
//   64 scattered 8-byte pieces per instruction and was bound by L2 write requests, 134 M per GiB).
// Phasors: per-lane FP64 state advanced by one FP64 complex multiply per tile (channel 0's NCO at
// the lane's first staged sample; channel c's correction at the tile's first output, lanes 0..63),
// set up with one sincospi per lane per launch.  Within a tile the correction angle grows by
// 64*ddelta_c per output (<= 1e-2 rad over the tile): a 3-term FP32 series.
#include "chan.hip.h"
#include "cfft.hip.h"

namespace qk {

constexpr int kChK = 64;                 // channels == branches == decimation
constexpr int kChT = 64;                 // output times per tile
constexpr int kChRowX = 68;              // staged input: 64 samples per row + 4 pad (conflict-free branch reads)
constexpr int kChRow3 = 65;              // output tile [channel][n'] row pitch

__device__ __forceinline__ float2 rot_gain(float2 x, double2 p, long long g, float gm1) {
    const float gain = fmaf((float)(int)(g & 511), gm1, 1.0f);
    const float pr = (float)p.x * gain, pi = (float)p.y * gain;
    return make_float2(fmaf(x.x, pr, -x.y * pi), fmaf(x.x, pi, x.y * pr));
}

template <bool INV>   // INV: channel c sits at +c/64 turn per sample relative to channel 0, else -c/64
__global__ __launch_bounds__(256, 3) void chan_uniform_kernel(const ChanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const int P = a.P, Q = a.Q;
    const float2* __restrict__ in = a.in;

    if ((int)blockIdx.x == a.nwg) {
        // shared history: last P samples of hist ++ in, rotated by channel 0's NCO
        for (int i = t; i < P; i += 256) {
            const long long g = a.count - P + i;
            float2 v;
            if (g < 0) {
                v = a.hist[g + P];
            } else {
                v = rot_gain(in[g], fx_phasor(a.phase0 + (unsigned long long)g * a.dphase0), g, 0.0f);
            }
            a.hist_next[i] = v;
        }
        return;
    }

    // ---- per-lane constants -------------------------------------------------------------------
    const int nl = t >> 2, sub = t & 3;          // roles in the branch / radix-16 phase
    const int s = P & 63;
    // taps: 1 KB LDS table behind the tile buffers (lanes of a wave read 4 distinct words: broadcast)
    float* tapl = reinterpret_cast<float*>(lds + a.lds_elems);
    tapl[t] = a.taps[t];
    float2 tw[16];                               // exp(+-j 2pi c0 sub / 64)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        tw[i] = a.tw64[(i * sub) & 63];          // exp(-j 2pi m / 64); conjugated below when INV
        if (INV) tw[i].y = -tw[i].y;
    }
    // store-phase tables behind the taps: corrA[64] (per tile), {theta_c, gm1_c}[64] (per launch)
    float2* corrA = reinterpret_cast<float2*>(tapl + 256);
    float2* cst = corrA + 64;
    double2 corr = make_double2(1.0, 0.0), corr_step = corr;
    if (t < 64) {
        const long long ddel_c = a.ddelta[t];            // dphase_c - dphase_0 -+ c*2^58 (tiny, signed)
        const long long jc0 = (long long)blockIdx.x * (kChT * kChK) - P + a.kcentre;   // window centre of the first output
        corr = fx_phasor(a.dphi[t] + (unsigned long long)(jc0 * ddel_c));
        corr_step = fx_phasor((unsigned long long)(ddel_c * (long long)(kChT * kChK) * (long long)a.nwg));
        cst[t] = make_float2((float)((double)(ddel_c * (long long)kChK) * 3.4061215800865545e-19), a.gm1[t]);   // 2pi / 2^64
    }
    const int wv = t >> 6, ln = t & 63;              // roles in the store phase: channel group, output time
    double2 ph0 = fx_phasor(a.phase0 + (unsigned long long)((long long)blockIdx.x * (kChT * kChK) - P + t) * a.dphase0);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.nwg) {
        const long long n0 = (long long)tile * kChT;       // first output time of the tile
        const long long jb = n0 * kChK - P;                 // stream position of staged element 0
        const int span = (kChT - 1 + Q) * 64;
        // ---- stage xr ---------------------------------------------------------------------
        {
            double2 ph = ph0;
            ph0 = dcmul(ph0, a.rot_tile);
            for (int u = t; u < span; u += 256) {
                const long long g = jb + u;
                float2 v = make_float2(0.0f, 0.0f);
                if (g < 0) {
                    if (g + P >= 0) v = a.hist[g + P];
                } else if (g < a.count) {
                    v = rot_gain(in[g], ph, g, 0.0f);
                }
                lds[(u >> 6) * kChRowX + (u & 63)] = v;
                ph = dcmul(ph, a.rot256);
            }
        }
        __syncthreads();
        // ---- branch sums U[mu = sub + 4i], then radix-16 over i ---------------------------------
        float2 U[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int p = (sub + 4 * i + s) & 63;
            float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < Q) {
                    const float2 x = lds[(nl + q) * kChRowX + p];
                    const float h = tapl[64 * q + p];
                    acc.x = fmaf(h, x.x, acc.x);
                    acc.y = fmaf(h, x.y, acc.y);
                }
            }
            U[i] = acc;
        }
        fft16<INV>(U);                               // over i -> c0 at U[rev16(c0)]
        __syncthreads();                             // everyone is done reading xr
        // layout 2: row (n', g = c0 >> 2), 17 pitch, element (c0 & 3)*4 + sub
#pragma unroll
        for (int c0 = 0; c0 < 16; c0++) {
            const float2 v = (c0 == 0 || sub == 0) ? U[rev16(c0)] : cmulc<false>(U[rev16(c0)], tw[c0]);
            lds[(nl * 4 + (c0 >> 2)) * 17 + (c0 & 3) * 4 + sub] = v;
        }
        __syncthreads();
        // ---- radix-4 across sub: lane (n' = t>>2, g = t&3) owns c0 = 4g .. 4g+3 -----------------
        float2 R[16];
#pragma unroll
        for (int k = 0; k < 16; k++) R[k] = lds[t * 17 + k];
#pragma unroll
        for (int c0l = 0; c0l < 4; c0l++) fft4<INV>(R[4 * c0l], R[4 * c0l + 1], R[4 * c0l + 2], R[4 * c0l + 3]);
        __syncthreads();
        // layout 3: [channel c = 16*c1 + 4g + c0l][n'], pitch 65
        {
            const int g = t & 3;
#pragma unroll
            for (int c0l = 0; c0l < 4; c0l++)
#pragma unroll
                for (int c1 = 0; c1 < 4; c1++) lds[(16 * c1 + 4 * g + c0l) * kChRow3 + nl] = R[4 * c0l + c1];
        }
        if (t < 64) {
            corrA[t] = make_float2((float)corr.x, (float)corr.y);
            corr = dcmul(corr, corr_step);
        }
        __syncthreads();
        // ---- per-channel correction and contiguous stores: wave wv, lane n' = ln ----------------------
        {
            const long long nn = n0 + ln;
            const long long j = nn * kChK - P + a.kcentre;                   // window-centre position of this output
            const float jm = (float)(int)(j & 511);
            const float fl = (float)ln;
            float2* __restrict__ o = a.out + nn;
            const bool live = nn < a.nout;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int c = wv * 16 + k;
                const float2 y = lds[c * kChRow3 + ln];
                const float2 A = corrA[c], tg = cst[c];
                const float ang = fl * tg.x, a2 = ang * ang;
                const float br = fmaf(a2, -0.5f, 1.0f), bi = fmaf(a2 * ang, -1.0f / 6.0f, ang);
                const float gain = fmaf(jm, tg.y, 1.0f);
                const float pr = fmaf(A.x, br, -A.y * bi) * gain, pi = fmaf(A.x, bi, A.y * br) * gain;
                const float2 r = make_float2(fmaf(y.x, pr, -y.y * pi), fmaf(y.x, pi, y.y * pr));
                if (live) o[(size_t)c * a.out_stride] = r;
            }
        }
        __syncthreads();   // layout 3 is read before the next tile's staging overwrites it
    }
}

int launch_chan_uniform(const ChanArgs& a, int grid, size_t lds_bytes, hipStream_t stream) {
    if (a.inv) hipLaunchKernelGGL(chan_uniform_kernel<true>, dim3(grid), dim3(256), lds_bytes, stream, a);
    else hipLaunchKernelGGL(chan_uniform_kernel<false>, dim3(grid), dim3(256), lds_bytes, stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace qk
