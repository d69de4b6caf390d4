// pfb_dec.hip.h -- polyphase overlap-save decimator (decimate by 8), one WAVE per segment (gfx950).
//
// PolyphaseResampler<complex_t> with interp 1, decim 8 (src/dsp/resampling.h:99-132) and the fused VFO
// (src/dsp/vfo.h:19-36) on large calls.  Same operator, same state and boundary as fir_fft_dec_kernel
// (fft_fir.hip.h); what changes is the factorisation of the fast convolution:
//
//   y[n'] = sum_k h[k] s[8 n' - P + k]              (P = ntaps: the resampler's window ends one sample early)
//         = sum_{c<8} (gamma_c * u_c)[n']            u_c[a] = seg[8a + c]   the 8 polyphase columns of the input
//                                                    gamma_c[q] = g[8q + 7 - c], g[j] = h[N-1-j]   32 taps each at N = 256
//
// so a segment of 4096 input samples is EIGHT 512-point transforms of the decimated columns (9 radix-2 stages' worth
// instead of the 12 of one 4096-point transform), the eight spectra are multiplied by the column filters' spectra and
// SUMMED, and ONE 512-point inverse yields 513 - Q valid outputs (Q = ceil(N/8) taps per column; 481 at 256 taps).
//
// Execution model: a wave owns a whole segment -- 64 lanes x 64 complex values in 128 VGPRs -- and never meets
// another wave: no workgroup barrier anywhere in the segment loop (fir_fft_dec_kernel: eight per segment at two
// workgroups per CU).  Loads are whole 512-byte rows (lane = sample within the row).  The first two radix-8 passes of
// every column run in registers without any exchange (the lane holds a column's samples a = 64 j + 8 q' + g' for all
// (j, q'): pass 1 over j, compile-time twiddles W64^(q' k0), pass 2 over q'); ONE wave-private LDS exchange (eight
// rounds of 8 values per lane, 5 KB) regroups the eight lanes g' for the last pass; the spectrum products are summed
// over the eight columns with a reduce-scatter on DPP row operations; three small exchanges carry the 512-point
// inverse.  Tables (column spectra 40 KB, twiddles) sit in LDS, laid out so every wide read is conflict-free.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int kPfbD = 8;          // decimation = number of polyphase columns
constexpr int kPfbF = 512;        // transform length per column
constexpr int kPfbSeg = 4096;     // input samples per segment
constexpr int kPfbNT = 256;       // 4 waves per workgroup, each on its own segments
constexpr int kPfbRow = 10;       // LDS / table row pitch in complex values (80 B: 16-byte aligned, conflict-free, see pfb_dec.hip)
constexpr int kPfbMaxQ = 136;     // taps per column the dispatch accepts (>= 377 valid outputs per 512; 1024 taps at decimation 4 need 129)

struct PfbArgs {
    const float2* in;
    float2* out;
    const float2* hist;           // H raw samples preceding in[0] (fused VFO: de-rotated, as for fir_fft_dec_kernel)
    float2* hist_raw_next;
    const float2* hist_keep;
    float2* hist_next;
    const float2* tables;         // G [512][10] | TW [64][10] | TI1 [64][10] | TI2 [8][10] | EL [64]   (host layout == LDS layout)
    long long count, nout;
    int H;                        // history length = taps per phase = ntaps
    int Q;                        // taps per column = ceil(ntaps / 8)
    int Lo;                       // valid outputs per segment = 513 - Q
    int nseg;                     // segments of this call = ceil(nout / Lo)
    int nwg;                      // workgroups doing segments (grid = nwg + 1: the last one hands over the history)
    int rot;
    unsigned long long phase_in0, phase0, dphase;
    double2 rot_step;             // exp(j 2pi * 8 Lo * (4 nwg) * dphase): a wave's step from one of its segments to the next
    double2 pb_base;              // exp(j 2pi (phase0 + (8 (-(Q-1)) - 1) dphase)): phase of segment 0's element a' = 0
    double2 seg_pow[12];          // exp(j 2pi * 8 Lo * 2^k * dphase): segment b's base = pb_base * prod over the set bits of b (no sincos on the device)
    float2 wtab[8];               // exp(j 2pi * 512 b1 * dphase)
    float gm1;
    int PH;                       // output phases per column output: 1 = decimate by 8, 2 = decimate by 4 (pfb_dec.hip)
    int real;                     // 1: real data (in / out / hist are float arrays; two segments per transform, pfb_dec.hip)
};

constexpr int kPfbTableElems = (512 + 64 + 64 + 8) * kPfbRow + 64;   // + EL: exp(j 2pi 8 l dphase) per lane (fused VFO)
constexpr int kPfbG1Elems = 512 * 8;                                 // decimate by 4: the odd outputs' column spectra, rows of 8 (chunk-swizzled)

int launch_pfb_dec(const PfbArgs& a, hipStream_t stream);

}  // namespace qk
