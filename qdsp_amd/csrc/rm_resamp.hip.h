// rm_resamp.hip.h -- rational polyphase resampler (interp L > 1) as an FP32 matrix product on the MFMA units (gfx950).
//
// PolyphaseResampler<complex_t>::run (src/dsp/resampling.h:99-132) walks  out[n] = sum_t tapPhases[p_n][t] * buf[o_n + t],
// o_n = (n M) / L, p_n = (n M) % L, buf = history (P = taps per phase) ++ input.  With n = L b + i the offset splits into
// o_n = M b + off_i (off_i = (i M) / L < M): the L outputs of PERIOD b read the M + P - 1 samples from buf[M b] on, output i
// through the fixed row  W[i][c] = tapPhases[(i M) % L][c - off_i]  (zero outside the P-wide band):
//
//     Y[i][b] = sum_c W[i][c] * X[c][b],      X[c][b] = buf[M b + c]
//
// -- one L x (M + P - 1) banded matrix applied to every period.  v_mfma_f32_4x4x1_16B_f32 multiplies sixteen independent
// 4 x 1 by 1 x 4 blocks per instruction: block = four consecutive outputs i (its own band: 3 M / L + P columns, 20 for
// 48 kHz <-> 44.1 kHz with 16 taps per phase), the four columns of the result = four consecutive periods, one band
// column per instruction.  A lane ends up with four consecutive outputs of one period -- 32 bytes -- and the sixteen
// blocks of a group cover 64 consecutive outputs, so the results go to memory as they are, no transposition.  (A
// period of at most 8 blocks shares the instruction with further period quads: block = (period quad, output block).
// Short periods are first merged J at a time -- the same resampler with L' = J L, M' = J M -- until a row holds its
// band.)
// The FP32 MFMA is an fmaf chain (one rounding per product): the numerics of the direct kernel; columns beyond a row's
// band meet zero taps.
//
// Execution model (as decim_mfma_kernel): a wave owns its tiles and never meets another wave.  A tile is 4 G periods
// (4 G M samples: G so that it is 500-700 samples); its samples are loaded linearly one tile ahead into registers,
// rotated (fused VFO) and written to the wave's LDS rows, one period per row, each row extended by the first `ext`
// samples of the next period so that no band wraps; the A operands (the band matrix in lane order, built on the host)
// sit in LDS, shared by the workgroup's four waves; the B operand of a step is one ds_read_b64 (re and im product).
// resamp_any_kernel, which this replaces where it applies, reads every staged sample back from LDS once per tap and
// output (48 kHz -> 44.1 kHz: 0.36 ms per 2^26 input samples, LDS-bound).
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int kRmNE = 12;         // samples per lane and tile held in registers: 4 G M + ext <= 768
constexpr int kRmMaxGrp = 3;      // groups of 16 blocks = 64 outputs per period: L <= 192
constexpr int kRmMaxKB = 40;      // band columns per block

struct RmArgs {
    const float2* in;
    float2* out;
    const float2* hist;           // P samples preceding in[0] (fused VFO: rotated)
    float2* hist_next;
    const float* atab;            // [ngrp][KB][64] A operands, then [ngrp][64] ints: first band column of the lane's block, then
                                  // [ngrp][64] ints: (period quad of the lane's block within a step << 16) | output block (0xffff: idle)
    long long count, nout;
    int L, M, P;
    unsigned minv;                // ceil(2^32 / M)
    int ngrp, KB;                 // groups of 16 blocks, band columns per block
    int G;                        // period quads per tile (a multiple of qpb)
    int qpb;                      // period quads per MFMA step: 1, or 16 / (blocks per period) when a period has <= 8 blocks
    int ext;                      // columns of the next period appended to each row
    int pitch;                    // row pitch in samples (>= M + ext)
    int total;                    // samples staged per tile = 4 G M + ext
    int ntiles;                   // tiles of 4 G periods
    int nwaves;                   // waves looping over tiles (grid = ceil(nwaves / 4) + 1: the last workgroup hands over the history)
    unsigned long long phase0, dphase;
    double2 rot_step;             // exp(j 2pi nwaves 4 G M dphase): a wave's step from one of its tiles to the next
    float2 rot_k[kRmNE];          // exp(j 2pi 64 e dphase)
    float gm1;
};

inline size_t rm_lds_bytes(int ngrp, int KB, int G, int pitch, bool real = false) {
    return (size_t)((ngrp * KB * 64 + 2 * ngrp * 64 + 3) & ~3) * 4 + 4 * ((size_t)4 * G * pitch + 64) * (real ? 4 : 8);      // A operands + block tables + four waves' tiles
}

int launch_rm_resamp(const RmArgs& a, bool rot, bool real, hipStream_t stream);      // real: float samples (in / out / hist reinterpreted), never with rot

}  // namespace qk
