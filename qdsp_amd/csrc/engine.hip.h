// engine.hip.h -- what the translation units of the host side share (round 3: qdsp_hip.hip, 3 159 lines, was split by operator):
//   qdsp_hip.hip   the engine behind FIR / resampler / xlator / fused VFO (plans, tables, kernel selection: process_dev) and
//                  their C entry points
//   chan_ops.hip   the channelizer (Splitter -> N x VFO as one operator): uniform polyphase plan, batched per-channel kernels,
//                  qdsp_hip_chan_cf32_*
//   misc_ops.hip   element-wise math blocks (src/dsp/math.h), synthetic IQ, events, the timing / introspection helpers
// Everything here lives in namespace qh (internal: nothing of it is declared in include/qdsp_hip.h).
#pragma once
#include "../../include/qdsp_hip.h"
#include "kernels.hip.h"
#include "fft_fir.hip.h"
#include "chan.hip.h"
#include "pfb_dec.hip.h"
#include "mf_dec.hip.h"
#include "rm_resamp.hip.h"
#include "fir_lat.hip.h"
#include "knobs.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

namespace qh {


#define HIPCHK(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return -(int)e_;         \
    } while (0)

constexpr int kMaxDynLds = 64 * 1024;  // default dynamic-LDS ceiling; tiles are sized under it

enum Kind : int { KIND_FIR = 1, KIND_DECIM = 2, KIND_XLATE = 3, KIND_VFO = 4, KIND_CHAN = 5, KIND_SINE = 6 };
constexpr uint32_t kMagic = 0x51445350u;  // "QDSP"

struct Launch {
    const char* name = "";
    int grid = 0, block = 0, lds = 0;
};

// One engine serves FIR, resampler, xlator and the fused VFO: they differ only in
// (ch, interp, decim, rotate) and in which kernel the launch picks.
struct Engine {
    uint32_t magic = kMagic;
    Kind kind;
    int device = 0;
    int ch = 2;               // floats per sample
    int L = 1, M = 1;         // interp, decim
    int ntaps = 0;            // prototype length
    int P = 0;                // taps per phase = ceil(ntaps / L)
    int H = 0;                // history length in samples
    bool rotate = false;
    bool has_filter = true;
    // NCO: fixed-point turns, 2^64 == one turn
    unsigned long long phase = 0, dphase = 0;
    long double dturns = 0.0L;
    float inc_re = 1.0f, inc_im = 0.0f;
    bool volk_gain = true;     // emulate the VOLK rotator's magnitude sawtooth (see rotate())
    float gm1 = 0.0f;          // |phase_inc| - 1
    // device state
    float* d_taps = nullptr;
    double2* d_nco_tab = nullptr;  // tile_phasor tables of the direct kernels (fused NCO)
    unsigned long long nco_key_dphase = 0;
    long long nco_key_S = 0;
    int nco_key_NT = 0, nco_key_na = 0;
    // overlap-save VFO: the history un-rotated (the kernels filter raw samples), double-buffered like d_hist.
    // raw_valid: d_hist_raw[cur] matches d_hist[cur] (left there by the previous overlap-save call's hand-over);
    // anything else that touches the history or the NCO clears it and the next call de-rotates d_hist[cur] once.
    float* d_hist_raw[2] = {nullptr, nullptr};
    int hist_raw_cap = 0;
    bool raw_valid = false;
    float* d_taps_lm = nullptr;  // resamp_lm_kernel's per-sub-filter branch-major taps (small interp only)
    size_t taps_lm_t_off = 0;    // offset (floats) of the transposed copy used when decim == 1    // core layout (branch-major) or phases [L][P]
    float* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    hipStream_t last_stream = nullptr;   // process_ex / generate: the stream of the previous call (its own or the shared one)
    hipEvent_t done_ev = nullptr;        // QDSP_HIP_LINK_HOST_DEFERRED: recorded behind the call's work instead of waiting for it
    // Host input of a block-graph call (process_ex, round 4): the upload runs on its OWN stream, ordered behind the previous call's kernel (which read
    // the staging buffer) and in front of this call's kernel by events -- so block k + 1's host-to-device copy overlaps block k's device-to-host copy
    // (two DMA directions, two streams) and the call returns as soon as the input buffer has been read.
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up = nullptr, ev_kernel = nullptr;
    bool kernel_recorded = false;
    size_t hist_cap = 0;        // samples
    // host-pointer path
    hipStream_t stream = nullptr;
    void* d_in = nullptr;
    void* d_out = nullptr;
    int max_block = 0;
    size_t out_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // overlap-save fast convolution (FIR<complex_t> with many taps), fft_fir.hip.h
    int fir_mode = 0;           // 0 auto, 1 direct form, 2 overlap-save FFT
    int auto_veto = 0, auto_mode = 0;   // integer decimators / fused VFO, AUTO: per-call exceptions to the rule chain named by the measured table (decim_table.inc)
    int auto_pick = 0;          // FIR<complex_t>, AUTO: the kernel family the measured table names for this call (dispatch_table.inc), 0 = the rule chain
    float2* d_fft_H = nullptr;  // spectrum of the reversed taps / F, digit-reversed
    float2* d_fft_TA = nullptr;
    float2* d_fft_TB = nullptr;
    int fft_ntaps = -1;         // tap count d_fft_H was built for (-1: not built)
    bool fft_tw_ready = false, f1k_tw_ready = false;   // twiddle tables allocated AND uploaded (set last: a failed first call retries them)
    // 1024-point segments, one wave each (fft1k_fir.hip): spectrum in that kernel's pass-C order + its twiddles
    float2* d_f1k_H = nullptr;
    float2* d_f1k_T = nullptr;  // [16][64] W1024^(l ka), then [16][4] W64^(j kb1)
    int f1k_ntaps = -1;
    unsigned long long f1k_dphase = 0;
    // per-call NCO constants of the overlap-save launches, kept while the increment stands (16 long-double sincos
    // per call are ~3 us of host time: more than a reference-sized call's kernel)
    float2 wtab1k[16], wtab4k[16];
    unsigned long long wtab1k_dphase = 0, wtab4k_dphase = 0, rot_step_dphase = 0;
    bool wtab1k_ok = false, wtab4k_ok = false, rot_step_ok = false;
    long double rot_step_mult = 0.0L;
    double2 rot_step_val;
    unsigned long long fft_dphase = 0;   // NCO increment d_fft_H was built for (fused VFO), 0 otherwise
    float* d_taps_rm = nullptr;    // rational MFMA resampler (rm_resamp.hip.h): A operands + first columns, built with the taps
    int rm_ngrp = 0, rm_KB = 0, rm_ext = 0, rm_pitch = 0, rm_G = 0, rm_J = 1, rm_qpb = 1;
    bool rm_big_only = false;     // plan admitted by the round-3 extension of the rule: chip-filling calls only (rm_min_count)
    float* d_taps_mf = nullptr;    // MFMA decimator (mf_dec.hip.h): [2 KJ][64] A operands, built with the taps
    int mf_KJ = 0, mf_QS = 1, mf_keep2 = 0;
    // polyphase overlap-save decimate-by-8 (pfb_dec.hip.h): column spectra + twiddles, built for (pfb_ntaps, pfb_dphase)
    float2* d_pfb = nullptr;
    int pfb_ntaps = -1;
    int pfb_M = -1;               // (decimate by 8, or by 4 as two output phases: another table)
    unsigned long long pfb_dphase = 0;
    std::vector<float> taps_host;
    // A retune (set_phase_inc) may come from a control thread while the worker is inside process*: the new
    // increment is staged here as the two float bit patterns (0 = nothing staged: a zero increment is refused)
    // and applied by whoever next enters a call that reads the NCO state (apply_pending_inc).
    std::atomic<unsigned long long> pending_inc{0};
    // tuning / introspection
    int R = 0, NT = 0;          // 0 = pick automatically
    Launch last;
};

struct AnyPlan { int Pp, tap_bytes; bool pad; long long tile, span; int ks_lanes, ks_shift, ks_chunk; };

constexpr uint32_t kChanMagic = 0x4348414eu;  // "CHAN"
// the channelizer handle (chan_ops.hip; the harness helpers of misc_ops.hip take either kind of handle)
struct Chan {
    uint32_t magic = kChanMagic;
    int device = 0;
    int nchan = 0;
    std::vector<Engine*> vfo;      // one fused xlate+FIR+decimate engine per channel
    hipStream_t stream = nullptr;  // host-pointer path
    void* d_in = nullptr;
    void* d_out = nullptr;
    int max_block = 0;
    size_t d_in_cap = 0;           // samples d_in holds (>= max_block; grown by process_links for host inputs)
    size_t out_cap = 0;            // samples per channel
    // uniform polyphase fast path (chan.hip): 64 channels spaced +-1/64 turn/sample, decim 64
    int mode = 0;                  // QDSP_HIP_FIR_AUTO / _DIRECT (one fused VFO kernel per channel) / _FFT (= fast path if the plan allows)
    bool volk_gain = true;
    int ntaps = 0, interp = 1, decim = 1;
    float2* d_taps = nullptr;      // prototype taps * exp(j k dphase_0), padded to 256 (rebuilt when dphase_0 changes)
    unsigned long long gt_dphase = 0;
    bool gt_valid = false;
    float2* d_tw64 = nullptr;
    float* d_hist[2] = {nullptr, nullptr};   // P raw input samples (shared by all channels)
    int cur = 0;
    // batched per-channel form (resamp_any_batch_kernel): the prototype's [L][P] phase table and the per-channel
    // constants, re-uploaded when a channel is retuned or its buffers change (batch_key: what the table was built from)
    float* d_phases = nullptr;
    qk::AnyChanConst* d_batch = nullptr;
    std::vector<qk::AnyChanConst> batch_key;
    // the same for the MFMA decimator (decim_mfma_batch_kernel: large integer decimations)
    qk::MfChanConst* d_batch_mf = nullptr;
    std::vector<qk::MfChanConst> batch_mf_key;
    Launch last;
};
inline Chan* as_chan(void* h) {
    Chan* c = static_cast<Chan*>(h);
    return (c && c->magic == kChanMagic) ? c : nullptr;
}

// ---- shared by the translation units (defined in qdsp_hip.hip) ----
Engine* as_engine(void* h, Kind k);
hipError_t wait_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t ev, hipStream_t s);
hipStream_t shared_stream(int device);
long double turns_of(float re, float im);
unsigned long long fx_of_turns(long double t);
void unit_of_fx(unsigned long long ph, long double mult, double* c, double* s);
void unit_of_fx_c(unsigned long long ph, long double mult, double* c, double* s);
int64_t out_size(const Engine* e, int64_t count);
int win_R(int M, int P);
bool use_win(const Engine* e);
bool use_lm(const Engine* e);
bool mf_plan(const Engine* e, int* KJ, int* QS, int* keep2);
int upload_taps(Engine* e, const float* taps, int ntaps);
int configure(Engine* e, const float* taps, int ntaps, int interp, int decim);
void set_inc_now(Engine* e, float re, float im);
void set_inc(Engine* e, float re, float im);
void apply_pending_inc(Engine* e);
int ensure_io(Engine* e, int max_block);
int create(void** h, Kind kind, int device, int ch, bool rotate, bool has_filter, int max_block);
void destroy(Engine* e);
int nco_tables(Engine* e, long long S, int NT, int na, const double2** tab);
int launch_core_t(Engine* e, qk::CoreArgs& a, hipStream_t s);
AnyPlan any_plan(int L, int M, int P, int ch, long long nout = -1);
size_t fill_any_geometry(qk::AnyArgs& a, int ch, bool* lt, bool* pad, int nchan = 1);
int fft_dec(const Engine* e);
bool any_direct_wins(const Engine* e);
bool fft_eligible(const Engine* e, int64_t count);
void host_spectrum(const std::vector<long double>& gr, const std::vector<long double>& gi, int F, std::vector<double>& re, std::vector<double>& im);
int fft_prepare(Engine* e);
bool fft1k_eligible(const Engine* e, int64_t count);
int fft1k_prepare(Engine* e);
int launch_fft1k(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s);
bool pfb_eligible(const Engine* e, int64_t count);
int pfb_prepare(Engine* e);
int raw_history(Engine* e, hipStream_t s, const float2** hist, float2** hist_raw_next);
int launch_pfb(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s);
int launch_fft(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s);
int launch_xlate_raw(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, float gm1, hipStream_t s);
int launch_xlate(Engine* e, const void* d_in, int64_t count, void* d_out, hipStream_t s);
void mf_tasks(qk::MfArgs& a, int64_t nout, int nchan, bool rot, bool real = false);
void mf_rot_tables(unsigned long long dphase, int M, int KJ, double2* step, float2* rot_k);
bool fir_lat_eligible(const Engine* e, int64_t count);
int launch_fir_lat(Engine* e, const void* d_in, int64_t count, void* d_out, hipStream_t s);
int launch_rm(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s);
int launch_mf(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s);
bool lm_yields_to_any(const Engine* e, int64_t nout);
bool win_yields_to_fft1k(const Engine* e, int64_t count);
int64_t mf_min_count(const Engine* e);
int64_t rm_min_count(const Engine* e);
int64_t process_dev(Engine* e, const void* d_in, int64_t count, void* d_out, void* stream);
int64_t process_host(Engine* e, const float* in, int count, float* out);
void* mapped_host_ptr(void* p);
int64_t process_ex(Engine* e, const void* in, int in_dev, int count, void* out, int out_dev);
int reset(Engine* e);
int get_history(Engine* e, float* hist);
int set_history(Engine* e, const float* hist);
int set_history_dev(Engine* e, const void* d_hist, void* stream);
int get_phase(Engine* e, float* re, float* im);
int set_phase(Engine* e, float re, float im);
int time_process(Engine* e, const void* d_in, int64_t count, void* d_out, void* stream, int iters, float* ms);
Engine* any_engine(void* h);

bool use_core(const Engine* e);

// staging-side NCO constants of the direct-form kernels (used by qdsp_hip.hip and chan_ops.hip)
template <class ARGS> void fill_stage_rot(ARGS& a, int NT) {
    unit_of_fx_c(a.dphase, (long double)NT, &a.rot_nt.x, &a.rot_nt.y);
    unit_of_fx_c(a.dphase, (long double)(8 * NT), &a.rot_8nt.x, &a.rot_8nt.y);
    for (int k = 0; k < 8; k++) {
        double c, sn;
        unit_of_fx_c(a.dphase, (long double)(k * NT), &c, &sn);
        a.rot_k[k] = make_float2((float)c, (float)sn);
    }
}

// tile_phasor's tables (kernels.hip.h), cached per handle for (dphase, S, NT, na); rebuilt on a retune or a new geometry

template <class ARGS> int fill_stage_rot(Engine* e, ARGS& a, int NT, long long S, long long first, long long ntiles) {
    fill_stage_rot(a, NT);
    a.nco_tab = nullptr;
    if (!e->rotate || 0) return 0;
    const int na = (int)((ntiles + 255) / 256) + 1;
    if (na > 65536) return 0;
    int rc = nco_tables(e, S, NT, na, &a.nco_tab);
    if (rc) return rc;
    a.nco_na = e->nco_key_na;
    const unsigned long long ph = a.phase0 + (unsigned long long)first * a.dphase;
    unit_of_fx(ph, 1.0L, &a.nco_e0.x, &a.nco_e0.y);
    return 0;
}

}  // namespace qh
