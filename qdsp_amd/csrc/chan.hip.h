// chan.hip.h -- interface of the uniform polyphase channelizer kernel (chan.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

struct ChanArgs {
    const float2* in;          // count samples
    float2* out;               // channel-major: channel c at out + c*out_stride
    const float2* hist;        // P raw input samples preceding `in`
    float2* hist_next;
    const float2* gtaps;       // prototype taps * exp(j k dphase0), zero padded to 256
    const float2* tw64;        // exp(-j 2pi m / 64), m = 0..63
    long long count;           // input samples of this call
    long long nout;            // outputs per channel of this call (count / M)
    long long out_stride;      // samples between channel rows of `out`
    int P;                     // taps per phase == history length (== ntaps: interp is 1)
    int Q;                     // ceil(ntaps / 64) <= 4
    int M;                     // decimation: 64 (critically sampled) or 8 / 16 / 32 (oversampled)
    int ntiles;                // wave tiles of 16 output times: ceil(nout / 16)
    int nwg;                   // persistent workgroups of `waves` independent waves (grid = nwg + 1: the last hands over history)
    int waves;                 // waves per workgroup: 4 (256 threads)
    int kcentre;               // tap index the per-channel deviation is evaluated at ((ntaps-1)/2)
    int st4;                   // 1: `out` is 16-byte aligned and out_stride even: full tiles are written with 16-byte stores
    int abl;                   // diagnostic builds only (QDSP_HIP_CHAN_ABL): ablation mask, 0 = the product
    int quad;                  // 1: keep the second-order term of the per-output deviation rotation (15 max|theta_c| > 1e-4)
    int inv;                   // 1: channel spacing +1/64 turn/sample, 0: -1/64
    unsigned long long phase0, dphase0;   // channel 0's NCO (fixed point, 2^64 = one turn)
    unsigned long long dphi[64];   // phi_c - phi_0 at the first sample of this call
    long long ddelta[64];          // dphase_c - dphase_0 -+ c*2^58: deviation from the uniform plan
    float gm1[64];                 // |phase_inc_c| - 1 (VOLK magnitude sawtooth), 0 = off
};

size_t chan_uniform_lds_bytes(int waves);
int launch_chan_uniform(const ChanArgs& a, int grid, hipStream_t stream);

}  // namespace qk
