// fft_fir.hip.h -- overlap-save fast convolution for long complex FIRs (gfx950).
//
// Why: direct form costs 2*ntaps FMAs per complex sample (1024 FLOP at 256 taps against 16
// algorithmic bytes: FP32-compute-bound at ~30 % of the HBM roofline).  A
// 4096-point overlap-save block costs ~135 FLOP per sample, which puts the filter back
// under the HBM roof.  Same operator, same state, same boundary (FIR<complex_t>::run,
// src/dsp/filter.h:51-74); results differ from the k-ordered FP32 sum only by FP32 FFT
// rounding (~3e-7 RMS relative, tests/test_gpu_parity.py) -- inside the 1e-5 bar.
//
// One workgroup = 256 lanes = one 4096-point block at a time, persistent over blocks.
//   x (F samples, H = ntaps-1 of them overlap) --A--> --B--> --C--> * Hf --C'--> --B'--> --A'--> y
// F = 16*16*16: three radix-16 passes done in registers (16 points per lane), separated by
// LDS transposes; the inverse runs the same passes backwards, so no bit-reversal pass
// exists: Hf is stored in the digit-reversed order pass C leaves the spectrum in.
// Per-lane constants (pass twiddles, Hf slice) are loaded once per workgroup.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int kFftN = 4096;
constexpr int kFftNT = 256;
constexpr int kFftRow1 = 272;   // layout 1: [k0][n1*16 + n0], row padded 256 -> 272 elements
constexpr int kFftRow2 = 17;    // layout 2: [k0*16 + k1][n0], row padded 16 -> 17 elements
constexpr int kFftLdsElems = 16 * kFftRow1;  // 4352 >= 256*17 = 4352

struct FftArgs {
    const float2* in;
    float2* out;
    const float2* hist;       // H raw samples preceding in[0] (fused VFO: de-rotated by the caller from the rotated form the handle keeps)
    float2* hist_raw_next;    // fused VFO: where the hand-over also leaves the next call's history un-rotated (or nullptr)
    const float2* hist_keep;  // the same H samples in the form the handle keeps them (rotated for the fused VFO): source of the hand-over
    float2* hist_next;
    const float2* Hf;         // [256][16]: Hf[(k0*16+k1)*16 + k2] = FFT(taps reversed)[k0 + 16 k1 + 256 k2] / F
    const float2* TA;         // [256][16]: exp(-j 2pi t k / 4096)
    const float2* TB;         // [16][16] : exp(-j 2pi lo k / 256)
    long long count;          // input samples of this call
    long long nout;           // outputs of this call
    int H;                    // history length (ntaps-1 for the FIR, taps per phase for the resampler)
    int dec;                  // 1, 2, 4, 8, 16: decimation handled by pruning the inverse transform
    int decm;                 // dec == 1 only: keep every decm-th output of the full inverse
    unsigned decm_inv;        // floor(2^32 / decm) + 1: x / decm == (x * decm_inv) >> 32 for x < 2^32 / decm (0: divide)
    int strided;              // dec == 1 only: resampler semantics -- y[n'] sits at stream position n'*decm - 1 (0 = FIR: y[n] at n)
    int rot;                  // 1: fused VFO -- Hf holds the spectrum of taps * exp(j k dphase), outputs are rotated
    int ov;                   // leading invalid elements of a segment (multiple of dec)
    int seg_shift;            // segment b starts at stream position b*L - seg_shift
    int L;                    // stream positions covered per segment = 4096 - ov
    int nblocks;
    int nwg;                  // persistent workgroups (grid = nwg + 1; the last one hands over history)
    int vec;                  // 1: in/out 16-byte aligned and segments start on even samples -> float4 path
    int grouped;              // dec >= 4: 1 = fir_fft_dec_kernel (groups of dec segments), 0 = one segment per workgroup
    int m_shift;              // fir_fft1k_kernel, strided: log2(decm) when decm is a power of two <= 64 (a lane keeps all or none of its 16 elements), else -1
    int abl;                  // diagnostic builds of fir_fft_dmapk_kernel: ablation mask (0: the product)
    unsigned* stamps;         // diagnostic build of fir_fft_dmapk_kernel only: [grid][16] per-phase tick sums (nullptr: off)
    int nt;                   // fir_fft_dmapk_kernel: bit 0 = non-temporal DMA of the rows no neighbour re-reads, bit 1 = non-temporal stores
    int dma;                  // 1: fir_fft_dma_kernel (FIR<complex_t>, 16-byte aligned in / out, overlap <= 2048): segments arrive by LDS-DMA
    int real2;                // 1: real samples (4-byte in/out/hist); block b = real segments 2b (re) and 2b+1 (im)
    // NCO (rot only).  x[j] exp(j phi(j)) filtered by h == exp(j phi(p - (N-1))) * (x filtered by
    // h[k] exp(j k dphase)) at output position p: the input is never rotated, only the KEPT outputs are.
    unsigned long long phase_in0;   // phase of in[0] (history hand-over, de-rotation of the history)
    unsigned long long phase0;      // phase_in0 - (ntaps-1)*dphase: output at position p gets phase0 + p*dphase
    unsigned long long dphase;
    double2 rot_step;         // exp(j 2pi nwg*L*dphase): block b -> b + nwg (per-segment kernel); x dec for the grouped kernel
    float2 wtab[16];          // exp(j 2pi 256*n2*dphase)
    float gm1;                // |phase_inc| - 1 (VOLK magnitude sawtooth: a real scale of the input samples), 0 = off
};

// fft1k_fir.hip: 1024-point segments, one wave each (reference-sized calls).  Same FftArgs; Hf / TA / TB point at
// that kernel's own tables (entry-major: [16][64] spectrum in its pass-C order, [16][64] W1024^(l ka), [16][4] W64^(j kb1)),
// wtab[i] = exp(j 2pi 64 i dphase), nblocks segments = the grid minus the hand-over workgroup.
constexpr int kFft1kN = 1024;
constexpr int kFft1kPitch = 84;
int launch_fir_fft1k(const FftArgs& a, hipStream_t stream);

// Defined in fft_fir.hip (its own translation unit: built with -fno-slp-vectorize, see there).
int launch_fir_fft(const FftArgs& a, int grid, hipStream_t stream);

}  // namespace qk
