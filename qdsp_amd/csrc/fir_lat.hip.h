// fir_lat.hip.h -- FIR<complex_t> on reference-sized calls: the direct form arranged for LATENCY (gfx950).
//
// out[n] = sum_k h[k] * s[n - (N-1) + k], s = history ++ input (src/dsp/filter.h:63-67).  A call of <= ~3e5 samples does
// not fill the chip: 65 536 samples are 17 overlap-save segments (one workgroup each: the call lasts as long as one
// segment's eight-barrier critical path, ~5.5 us of kernel), and fir_core_kernel -- built for throughput: R outputs per
// lane, taps through s_load in the loop, eight waves per SIMD to hide both -- takes ~7.5 us of kernel at 256 taps however
// short the call, the scalar loads' latency exposed once per eight taps.  Here a wave takes 64 consecutive outputs, one
// per lane; its 64 + N - 1 samples and the taps sit in LDS; the tap loop is unrolled by eight with the next chunk's ten
// LDS reads (eight samples, two broadcast tap quads) in flight while this chunk's eight dependent v_pk_fma_f32 issue:
// the critical path is the FMA chain itself, N x ~8 cycles.  Accumulation is in tap order with one FMA per tap and no
// padding taps: bit-identical to the k-ordered fmaf chain, as fir_core_kernel, and a NaN / Inf sample stays inside the N
// windows that hold it.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

struct FirLatArgs {
    const float2* in;
    float2* out;
    const float2* hist;           // N - 1 samples preceding in[0]
    float2* hist_next;
    const float* taps;            // h[0..N)
    long long count;
    int N;
    int Np;                       // N rounded up to a multiple of 8
    int nwaves;                   // wave tiles of 64 outputs (grid = ceil(nwaves / 4) + 1: the last workgroup hands over the history)
};

inline size_t fir_lat_lds_bytes(int Np) { return (size_t)(Np + 8) * 4 + (size_t)4 * (64 + Np + 8) * 8; }

int launch_fir_lat(const FirLatArgs& a, hipStream_t stream);

}  // namespace qk
