// chan.hip.h -- interface of the uniform polyphase channelizer kernel (chan.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

struct ChanArgs {
    const float2* in;          // count samples
    float2* out;               // channel-major: channel c at out + c*out_stride
    const float2* hist;        // P samples, already rotated by channel 0's NCO
    float2* hist_next;
    const float* taps;         // prototype taps, zero padded to 256
    const float2* tw64;        // exp(-j 2pi m / 64), m = 0..63
    long long count;           // input samples of this call
    long long nout;            // outputs per channel of this call (count / 64)
    long long out_stride;      // samples between channel rows of `out`
    int P;                     // taps per phase == history length (== ntaps: interp is 1)
    int Q;                     // ceil(ntaps / 64) <= 4
    int ntiles;                // ceil(nout / 64)
    int nwg;                   // persistent workgroups (grid = nwg + 1: the last hands over history)
    int kcentre;               // tap index the per-channel correction is evaluated at ((ntaps-1)/2)
    int inv;                   // 1: channel spacing +1/64 turn/sample, 0: -1/64
    int lds_elems;             // float2 elements of the tile buffer (256-float tap table + 2 x 64 float2 behind it)
    unsigned long long phase0, dphase0;   // channel 0's NCO (fixed point, 2^64 = one turn)
    double2 rot256;            // exp(j 2pi 256 dphase0)
    double2 rot_tile;          // exp(j 2pi 4096 nwg dphase0): a workgroup's tile -> its next tile
    unsigned long long dphi[64];   // phi_c - phi_0 at the first sample of this call
    long long ddelta[64];          // dphase_c - dphase_0 -+ c*2^58: deviation from the uniform plan
    float gm1[64];                 // |phase_inc_c| - 1 (VOLK magnitude sawtooth), 0 = off
};

int launch_chan_uniform(const ChanArgs& a, int grid, size_t lds_bytes, hipStream_t stream);

}  // namespace qk
