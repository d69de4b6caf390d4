// mf_dec.hip.h -- large integer decimations (the VFO's everyday job: 2.4 Msps -> 48 kHz is decimate by 50 with the
// 401 taps vfo.h:26-33 designs) as an FP32 matrix product on the MFMA units (gfx950).
//
//   y[n] = sum_k h[k] s[M n - P + k],  k = M q + r   (PolyphaseResampler<complex_t>::run, interp 1:
//                                                      src/dsp/resampling.h:121-125; + the xlator for the VFO)
//        = sum_q Z[q][n + q],   Z[q][rho] = sum_{r<M} h[M q + r] * X[rho][r],   X[rho][r] = s[M rho + r - P]
//
// Z = Taps (Q x M) . X^T (M x rows) is a plain matrix product: per tile of 16 input rows (16 M samples) it is K/4
// v_mfma_f32_16x16x4_f32 per component (A = taps, 16 rows q of which Q <= 16 are used; B = data, one of the 16 rows per
// lane column; K = M rounded up to 8).  The FP32 MFMA is bit-for-bit an fmaf chain (one rounding per product), so the
// numerics are those of the direct kernels.  What the matrix unit buys: no per-tap LDS reads (resamp_any_kernel: every
// sample read back Q times, LDS 74 % busy) and no cross-lane reduction per output (a lane-per-column VALU kernel, built first and
// not kept, spent twelve DPP adds per output) -- a sample is written to LDS once and read once, as a B operand.
//
// Execution model: a wave owns a run of T outputs and never meets another wave (no workgroup barrier).  Per tile it
//   - takes the tile's 16 M samples from registers (loaded one tile AHEAD, linearly: 64 consecutive samples per
//     instruction, all lanes busy at any M), rotates them (fused VFO) and writes them to its private LDS rows,
//   - issues the loads of the next tile,
//   - reads them back as B operands (ds_read_b128: two columns x re/im of one row; row pitch K + 2 samples keeps the
//     16 rows on distinct banks) and runs the 4 K/8 MFMAs,
//   - sums the 16 x 16 result along its diagonals: lane group g holds rows q = 4g..4g+3 of Z in four registers and
//     the row index on its 16 lanes, so the shift by q is a DPP row shift by the register index plus a per-group shift
//     by 4g (row-masked DPP), then one sum over the four lane groups.  Partial sums that reach into the previous tile's
//     outputs are carried in a register.
//
// 17-32 taps per column: a second tap set (tap rows 16-31; its diagonal sums belong one tile further back, so a tile
// completes the outputs two tiles behind it).  Decimations 130-256 (even): the same kernel at M / 2 -- the decimator
// by M / 2 with the same taps -- storing every other output (keep2).
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int kMfMaxKJ = 16;      // K / 8: decimations up to 128
constexpr int kMfMaxQ = 32;       // taps per column: two sets of 16 rows of the A operand

struct MfArgs {
    const float2* in;
    float2* out;
    const float2* hist;           // P samples preceding in[0] (fused VFO: rotated, as every direct-form kernel keeps them)
    float2* hist_next;
    const float* tapk;            // [QS][2 KJ][64] A operands: tapk[s][2 jj + a][l] = h[M (16 s + l % 16) + 8 jj + 2 (l / 16) + a], 0 outside
    long long count, nout;
    int P, M;
    unsigned minv;                // ceil(2^32 / M): idx / M == umulhi(idx, minv) for the tile-relative indices (< 2^11)
    int keep2;                    // 1: M is half the call's decimation; every other output is stored (at n / 2)
    int T;                        // outputs per wave task (multiple of 16)
    int ntasks;                   // wave tasks (grid = ceil(ntasks / 4) + 1: the last workgroup hands over the history)
    unsigned long long phase0, dphase;
    double2 rot_step;             // exp(j 2pi 16 M dphase): one tile further
    float2 rot_k[2 * kMfMaxKJ];   // exp(j 2pi 64 i dphase): load i of a tile
    float gm1;
    int defer_store;              // 1: a tile's outputs are stored one tile later, behind the wait for the next tile's samples (chip-filling calls: mf_dec.hip)
};

// N channels of one filter design in one launch (blockIdx.y = channel): the non-uniform channelizer and the
// Splitter -> N x VFO bank (routing.h:47-57 + vfo.h:19-36).  Per-channel constants sit in a device table, what changes
// every call (NCO phase at the first sample, output pointer) rides in the kernel arguments -- as resamp_any_batch_kernel.
constexpr int kMfBatchMax = 128;
struct MfChanConst {
    const float2* hist[2];        // the channel's two history buffers; `cur` selects the one to read
    unsigned long long dphase;
    double2 rot_step;
    float2 rot_k[2 * kMfMaxKJ];
    float gm1;
    int pad_;
};
struct MfBatchArgs {
    MfArgs a;                     // shared part (in, taps, geometry); per-channel fields are patched in
    const MfChanConst* tab;       // [nchan]
    long long out_stride;         // complex samples between the channels' output rows
    int cur;                      // history parity of this call (all channels flip together)
    int use_ptrs;                 // 1: channel c writes to outs[c]
    unsigned long long phase0[kMfBatchMax];
    void* outs[kMfBatchMax];
};

// KJ = ceil(M / 8) in 2..16; returns -1 for other shapes
int launch_mf_dec_batch(const MfBatchArgs& b, int nchan, int KJ, int depth, hipStream_t stream);
// qs: tap sets of 16 rows (1, or 2 for 17-32 taps per column: one tile in flight)
int launch_mf_dec(const MfArgs& a, int KJ, bool rot, int depth, int qs, bool real, hipStream_t stream);      // real: float samples (in / out / hist reinterpreted), never with rot

}  // namespace qk
