/*
 * qdsp_hip.h -- C ABI of libqdsp_hip.so: qdsp's FIR / polyphase-resampler / NCO-mixer
 * hot path as hand-written HIP kernels for MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference (AlexandreRouma/qdsp) has no FFI of its
 * own: its blocks call VOLK from inside `run()`.  Each entry point below replaces one of
 * those call sites (cited per function, paths relative to the reference tree), and is
 * what a HIP-backed `dsp::FIR<T>::run()` etc. binds to -- see INTEGRATION.md and the
 * host-side mirror in qdsp_amd/host/dsp/.
 *
 * Conventions
 *   - plain C: pointers, sizes, opaque `void*` handles; no C++/torch types, no exceptions.
 *   - complex samples are interleaved {re, im} float pairs (dsp::complex_t,
 *     src/dsp/types.h:65-66); `count` is always in SAMPLES, not floats.
 *   - return value: 0 (or a non-negative count where stated) on success, negative on
 *     failure: -(hipError_t) for runtime errors, QDSP_HIP_E* below for argument errors.
 *     qdsp_hip_error_string() decodes either.
 *   - every handle is bound to one device and is NOT thread-safe; as in the reference,
 *     `run()` owns it on the block's worker thread and setters are called with the block
 *     stopped (src/dsp/block.h:108-120).
 *   - `*_process`      : host pointers (the stream's readBuf / writeBuf).  Synchronous:
 *                        on return the output is in host memory, so `out.swap()` may
 *                        follow immediately (src/dsp/filter.h:69).
 *   - `*_process_dev`  : device pointers + a hipStream_t (as void*).  Asynchronous on that
 *                        stream; state (history, NCO phase) is advanced in stream order.
 *                        in/out must not alias.  `stream` NULL = HIP's default stream.
 *                        All calls on one handle must be stream-ordered by the caller.
 *   - `*_process_ex`   : like `*_process`, but each side says where its buffer lives
 *                        (`in_on_device` / `out_on_device`): a block whose neighbour is
 *                        another HIP-backed block reads / writes the stream's device-resident
 *                        buffer and skips that PCIe copy.  Synchronous.
 *   - one `process*` call == one `run()` of the reference block: history carries over
 *     exactly as the reference's memmove does, and the resampler's phase counter restarts
 *     at 0 (src/dsp/resampling.h:114,121).
 */
#ifndef QDSP_HIP_H
#define QDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QDSP_HIP_ABI_VERSION 1

#define QDSP_HIP_EINVAL (-10001)  /* bad argument                                    */
#define QDSP_HIP_ENOMEM (-10002)  /* host allocation failed                          */
#define QDSP_HIP_ESIZE (-10003)   /* count exceeds the handle's max_block            */
#define QDSP_HIP_ENODEV (-10004)  /* no usable gfx950 device / device index invalid  */
#define QDSP_HIP_ERCCL (-10005)   /* RCCL not loadable, or an RCCL call failed (ring)  */

/* ---- library ------------------------------------------------------------------------- */
int qdsp_hip_abi_version(void);
const char* qdsp_hip_error_string(int code);
int qdsp_hip_device_count(int* count);
/* Name / gcnArchName of a device into caller buffers (may be NULL). */
int qdsp_hip_device_info(int device, char* name, int name_len, char* arch, int arch_len,
                         int* compute_units);
/* The QDSP_HIP_* tuning / experiment variables are read ONCE, when the library first needs one, into an
 * immutable table -- no getenv on the call path (the reference blocks hand over <= 1e6 samples per call, src/dsp/stream.h:7:
 * a call lasts 3-8 us).  A process that changes its environment afterwards (tests, tuning scripts) calls this to have the
 * table rebuilt; calls already running keep the table they started with.  Returns 0. */
int qdsp_hip_reload_env(void);

/* Link codes of the *_process_ex / sine generate `*_on_device` arguments (the device-resident companion of
 * dsp::stream<T>, src/dsp/stream.h:21-125): 0 = host buffer; 1 = device buffer, complete on entry / on return;
 * QDSP_HIP_LINK_PIPELINED = device buffer on a pipelined link: the call runs on the library's one in-order stream
 * per device and does not wait for a block it hands to such a link -- the consumer's call runs on the same stream,
 * behind it.  Valid when both ends launch before they swap / flush the stream<T> (the blocks of
 * qdsp_amd/host/dsp do).  A host input is always waited for (its buffer is released on return). */
#define QDSP_HIP_LINK_HOST 0
#define QDSP_HIP_LINK_DEVICE 1
#define QDSP_HIP_LINK_PIPELINED 2
/* host buffer (pinned), completion deferred to the consumer: the call records the handle's done event
 * (qdsp_hip_set_done_event) behind its work and returns; whoever reads the buffer waits for that event first
 * (dsp::stream<T>::read does, the event travels with the buffer through swap()).  Output side only. */
#define QDSP_HIP_LINK_HOST_DEFERRED 3
int qdsp_hip_event_create(int device, void** ev);
int qdsp_hip_event_destroy(void* ev);
int qdsp_hip_event_wait(void* ev);
int qdsp_hip_set_done_event(void* handle, void* ev);   /* any FIR / resampler / mixer / VFO / sine handle */

/* ---- memory helpers (for hosts that do not link the HIP runtime themselves) ----------- */
/* Pinned host memory: replaces volk_malloc for stream buffers (src/dsp/stream.h:25-26) so
 * readBuf/writeBuf are DMA-able without staging. */
int qdsp_hip_host_alloc(void** p, size_t bytes);
int qdsp_hip_host_free(void* p);
/* Pin / unpin memory the caller already owns (e.g. an unmodified reference stream). */
int qdsp_hip_host_register(void* p, size_t bytes);
int qdsp_hip_host_unregister(void* p);
int qdsp_hip_dev_alloc(int device, void** p, size_t bytes);
int qdsp_hip_dev_free(int device, void* p);
int qdsp_hip_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes);
int qdsp_hip_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes);
int qdsp_hip_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes);
/* the same for a block that forwards data between links (Splitter, src/dsp/routing.h:47-57): ordered behind a
 * QDSP_HIP_LINK_PIPELINED source, not waited for when the destination link is pipelined as well */
int qdsp_hip_memcpy_d2d_link(int device, void* d_dst, const void* d_src, size_t bytes, int in_link, int out_link);
int qdsp_hip_memcpy_d2h_link(int device, void* h_dst, const void* d_src, size_t bytes, int in_link);
int qdsp_hip_device_sync(int device);

/* ---- FIR<complex_t> : src/dsp/filter.h:51-74 ------------------------------------------ */
/* Replaces the loop of volk_32fc_32f_dot_prod_32fc calls (filter.h:63-67) plus the
 * memcpy-into-history (filter.h:55) and memmove (filter.h:71):
 *     y[n] = sum_{k<ntaps} taps[k] * x[n - (ntaps-1) + k]
 * History = the last ntaps-1 input samples, zero after create/reset (the reference leaves
 * it uninitialised, filter.h:28).  max_block bounds `count` of the host-pointer path only
 * (STREAM_BUFFER_SIZE, src/dsp/stream.h:7). */
int qdsp_hip_fir_cf32_create(void** h, int device, const float* taps, int ntaps, int max_block);
int qdsp_hip_fir_cf32_process(void* h, const float* in_iq, int count, float* out_iq);
int qdsp_hip_fir_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                  void* hip_stream);
int qdsp_hip_fir_cf32_process_ex(void* h, const void* in, int in_on_device, int count, void* out,
                                 int out_on_device);
/* FIR<T>::updateWindow (filter.h:43-49): new taps; history is kept (resized, newest
 * samples preserved) as the reference keeps its buffer. */
int qdsp_hip_fir_cf32_set_taps(void* h, const float* taps, int ntaps);
/* Algorithm choice.  QDSP_HIP_FIR_DIRECT: direct form, taps accumulated in order 0..ntaps-1
 * with one fused multiply-add each (bit-identical to a k-ordered fmaf chain).
 * QDSP_HIP_FIR_FFT: 4096-point overlap-save fast convolution (~135 FLOP/sample instead of
 * 4*ntaps; FP32 FFT rounding, ~3e-7 RMS relative to the direct form).  QDSP_HIP_FIR_AUTO
 * (default): FIR<complex_t> follows a measured table (qdsp_amd/csrc/dispatch_table.inc: call size x taps -> the fastest of the
 * latency direct form, the direct form, the one-wave 1024-point and the 4096-point overlap-save kernels); other data types:
 * FFT for >= 8 taps on calls of >= 65536 samples (it runs at copy speed whatever the
 * taps; measured crossover), direct form otherwise. */
#define QDSP_HIP_FIR_AUTO 0
#define QDSP_HIP_FIR_DIRECT 1
#define QDSP_HIP_FIR_FFT 2
int qdsp_hip_fir_cf32_set_mode(void* h, int mode);
int qdsp_hip_fir_cf32_reset(void* h);
/* History access for the multi-GPU halo: `nsamples` = ntaps-1.
 * get/set copy through host memory; history_dev exposes the device buffer that the NEXT
 * process call will read, so an RCCL recv can land in it directly. */
int qdsp_hip_fir_cf32_history_len(void* h);
int qdsp_hip_fir_cf32_get_history(void* h, float* hist_iq);
int qdsp_hip_fir_cf32_set_history(void* h, const float* hist_iq);
int qdsp_hip_fir_cf32_history_dev(void* h, void** d_hist);
/* Install the `history_len` INPUT samples that precede the next call (device memory, e.g. the
 * buffer an RCCL recv of the neighbour's tail just filled), asynchronously on `hip_stream`.
 * xlate_fir_decim_cf32 keeps its history rotated (as the reference resampler's buffer holds
 * the xlator's output): it rotates the samples itself, with the NCO phases they would have
 * had (set / advance the phase first). */
int qdsp_hip_fir_cf32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
void qdsp_hip_fir_cf32_destroy(void* h);

/* ---- FIR<float> : src/dsp/filter.h:58-62 (volk_32f_x2_dot_prod_32f) -------------------- */
int qdsp_hip_fir_f32_create(void** h, int device, const float* taps, int ntaps, int max_block);
int qdsp_hip_fir_f32_process(void* h, const float* in, int count, float* out);
int qdsp_hip_fir_f32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                 void* hip_stream);
int qdsp_hip_fir_f32_process_ex(void* h, const void* in, int in_on_device, int count, void* out,
                                int out_on_device);
int qdsp_hip_fir_f32_set_taps(void* h, const float* taps, int ntaps);
int qdsp_hip_fir_f32_set_mode(void* h, int mode); /* as for cf32; the overlap-save form carries two real
                                                    * segments per complex transform (auto: >= 96 taps) */
int qdsp_hip_fir_f32_reset(void* h);
int qdsp_hip_fir_f32_history_len(void* h);
int qdsp_hip_fir_f32_get_history(void* h, float* hist);
int qdsp_hip_fir_f32_set_history(void* h, const float* hist);
int qdsp_hip_fir_f32_history_dev(void* h, void** d_hist);
int qdsp_hip_fir_f32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
void qdsp_hip_fir_f32_destroy(void* h);

/* ---- PolyphaseResampler<complex_t> : src/dsp/resampling.h:99-132 ----------------------- */
/* Rational interp/decim resampler; a pure decimator when interp == 1.  `taps` is the
 * prototype exactly as the window hands it over (already scaled by interp,
 * resampling.h:34); the phase split of buildTapPhases (resampling.h:137-166) happens
 * inside.  Replaces the volk_32fc_32f_dot_prod_32fc loop (resampling.h:121-125):
 *     P = ceil(ntaps/interp);  outCount = count*interp/decim;
 *     y[n] = sum_{t<P} phase[(n*decim) % interp][t] * x[(n*decim)/interp - P + t]
 * History = last P input samples (zero after create/reset, resampling.h:39).
 * process* return outCount (>= 0) or a negative error. */
int qdsp_hip_decim_cf32_create(void** h, int device, const float* taps, int ntaps, int interp,
                               int decim, int max_block);
int qdsp_hip_decim_cf32_process(void* h, const float* in_iq, int count, float* out_iq);
int64_t qdsp_hip_decim_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                        void* hip_stream);
int qdsp_hip_decim_cf32_process_ex(void* h, const void* in, int in_on_device, int count, void* out,
                                   int out_on_device);
/* updateWindow / setInSampleRate / setOutSampleRate (resampling.h:53-93) all funnel here. */
int qdsp_hip_decim_cf32_configure(void* h, const float* taps, int ntaps, int interp, int decim);
int64_t qdsp_hip_decim_cf32_out_size(void* h, int64_t count); /* calcOutSize, :95-97 */
/* QDSP_HIP_FIR_AUTO / _DIRECT / _FFT as for the FIR.  The overlap-save path serves interp == 1
 * with any decimation >= 2 (decim in {2, 4, 8, 16}: pruned inverse transform; others: full
 * inverse, every decim-th output stored).  AUTO (measured crossovers): short and medium filters
 * (decim 2..8, 10, 12, 16; up to 150..256 taps) run a strided-window direct kernel; decimations from 9 on with
 * up to ~9 taps per unit of decimation (the VFO's usual shape) the general direct kernel, which streams the
 * input once (4-16 lanes share an output's taps once a tile holds fewer outputs than a workgroup has lanes:
 * partial sums are then added in chunk order, not tap order); longer filters the overlap-save path on calls of
 * >= 65536 samples; the rest a direct form.  QDSP_HIP_FIR_DIRECT always means a direct form (decim <= 8: the
 * de-interleaved, k-ordered kernel; above: the general kernel).  interp > 1 is direct form whatever the mode. */
int qdsp_hip_decim_cf32_set_mode(void* h, int mode);
int qdsp_hip_decim_cf32_reset(void* h);
int qdsp_hip_decim_cf32_history_len(void* h); /* = taps per phase */
int qdsp_hip_decim_cf32_get_history(void* h, float* hist_iq);
int qdsp_hip_decim_cf32_set_history(void* h, const float* hist_iq);
int qdsp_hip_decim_cf32_history_dev(void* h, void** d_hist);
int qdsp_hip_decim_cf32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
void qdsp_hip_decim_cf32_destroy(void* h);

/* ---- PolyphaseResampler<float> : src/dsp/resampling.h:113-119 -------------------------- */
int qdsp_hip_decim_f32_create(void** h, int device, const float* taps, int ntaps, int interp,
                              int decim, int max_block);
int qdsp_hip_decim_f32_process(void* h, const float* in, int count, float* out);
int64_t qdsp_hip_decim_f32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                       void* hip_stream);
int qdsp_hip_decim_f32_process_ex(void* h, const void* in, int in_on_device, int count, void* out,
                                  int out_on_device);
int qdsp_hip_decim_f32_configure(void* h, const float* taps, int ntaps, int interp, int decim);
int64_t qdsp_hip_decim_f32_out_size(void* h, int64_t count);
int qdsp_hip_decim_f32_set_mode(void* h, int mode); /* as for cf32 (interp 1; auto: >= 32 taps per branch) */
int qdsp_hip_decim_f32_reset(void* h);
int qdsp_hip_decim_f32_history_len(void* h);
int qdsp_hip_decim_f32_get_history(void* h, float* hist);
int qdsp_hip_decim_f32_set_history(void* h, const float* hist);
int qdsp_hip_decim_f32_history_dev(void* h, void** d_hist);
int qdsp_hip_decim_f32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
void qdsp_hip_decim_f32_destroy(void* h);

/* ---- FrequencyXlator<complex_t> : src/dsp/processing.h:55-70 --------------------------- */
/* Replaces volk_32fc_s32fc_x2_rotator_32fc (processing.h:64): y[n] = x[n] * phase_n,
 * phase_{n+1} = phase_n * phase_inc, phase_0 = (1,0) after create.  phase_inc is the
 * float pair the reference computes in init/setFrequency (processing.h:20,48).
 * The device NCO is not recursive: phase_n = phase_0 * exp(j*n*arg(phase_inc)) with a
 * 64-bit fixed-point phase accumulator, so it has none of the float phasor's drift
 * (DESIGN.md "NCO").  The carried phase is readable/writable as the reference's
 * lv_32fc_t `phase`. */
int qdsp_hip_xlate_cf32_create(void** h, int device, float phase_inc_re, float phase_inc_im,
                               int max_block);
int qdsp_hip_xlate_cf32_process(void* h, const float* in_iq, int count, float* out_iq);
int qdsp_hip_xlate_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                    void* hip_stream);
int qdsp_hip_xlate_cf32_process_ex(void* h, const void* in, int in_on_device, int count, void* out,
                                   int out_on_device);
int qdsp_hip_xlate_cf32_set_phase_inc(void* h, float phase_inc_re, float phase_inc_im);
int qdsp_hip_xlate_cf32_get_phase(void* h, float* phase_re, float* phase_im);
int qdsp_hip_xlate_cf32_set_phase(void* h, float phase_re, float phase_im);
/* Jump the NCO as if `nsamples` samples had been processed (multi-GPU chunk start,
 * no communication needed for the phase). */
int qdsp_hip_xlate_cf32_advance(void* h, int64_t nsamples);
/* VOLK's rotator renormalises its float phasor only every 512 samples and at the end of a
 * call, so what it multiplies by has magnitude |phase_inc|^(n mod 512) (n from the start of
 * the call; |phase_inc| of the rounded float pair is 1 +- ~3e-8).  on = 1 (default)
 * reproduces that deterministic gain so results sit within float rounding of the
 * reference; on = 0 gives the ideal unit-magnitude NCO. */
int qdsp_hip_xlate_cf32_set_volk_gain(void* h, int on);
void qdsp_hip_xlate_cf32_destroy(void* h);

/* ---- SineSource : src/dsp/source.h:5-71 ------------------------------------------------- */
/* The reference runs volk_32fc_s32fc_x2_rotator_32fc over a buffer of ones (source.h:55-59):
 * out[n] = phase_n, phase_{n+1} = phase_n * phase_inc, phase_0 = (1,0).  Same NCO as
 * xlate_cf32 with no input traffic.  generate(): `out` on the host (synchronous copy) or
 * device-resident (out_on_device = 1); generate_dev(): asynchronous on `hip_stream`. */
int qdsp_hip_sine_cf32_create(void** h, int device, float phase_inc_re, float phase_inc_im, int max_block);
int qdsp_hip_sine_cf32_generate(void* h, int count, void* out, int out_on_device);
int qdsp_hip_sine_cf32_generate_dev(void* h, int64_t count, void* d_out, void* hip_stream);
int qdsp_hip_sine_cf32_set_phase_inc(void* h, float phase_inc_re, float phase_inc_im);
int qdsp_hip_sine_cf32_get_phase(void* h, float* phase_re, float* phase_im);
int qdsp_hip_sine_cf32_set_volk_gain(void* h, int on);
void qdsp_hip_sine_cf32_destroy(void* h);

/* ---- VFO : src/dsp/vfo.h:19-36 (FrequencyXlator -> PolyphaseResampler), fused ---------- */
/* One kernel does what the reference runs as two blocks/threads with a stream hop between
 * them: rotate while staging into LDS, then the polyphase dot products.  Semantics are
 * those of xlate_cf32 followed by decim_cf32 (history holds ROTATED samples, exactly as
 * the reference resampler's buffer does).  Returns outCount or a negative error. */
int qdsp_hip_xlate_fir_decim_cf32_create(void** h, int device, const float* taps, int ntaps,
                                         int interp, int decim, float phase_inc_re,
                                         float phase_inc_im, int max_block);
int qdsp_hip_xlate_fir_decim_cf32_process(void* h, const float* in_iq, int count, float* out_iq);
int64_t qdsp_hip_xlate_fir_decim_cf32_process_dev(void* h, const void* d_in, int64_t count,
                                                  void* d_out, void* hip_stream);
int qdsp_hip_xlate_fir_decim_cf32_process_ex(void* h, const void* in, int in_on_device, int count,
                                             void* out, int out_on_device);
int qdsp_hip_xlate_fir_decim_cf32_configure(void* h, const float* taps, int ntaps, int interp,
                                            int decim);
int qdsp_hip_xlate_fir_decim_cf32_set_phase_inc(void* h, float phase_inc_re, float phase_inc_im);
int qdsp_hip_xlate_fir_decim_cf32_get_phase(void* h, float* phase_re, float* phase_im);
int qdsp_hip_xlate_fir_decim_cf32_set_phase(void* h, float phase_re, float phase_im);
int qdsp_hip_xlate_fir_decim_cf32_advance(void* h, int64_t nsamples);
int qdsp_hip_xlate_fir_decim_cf32_set_volk_gain(void* h, int on);
int64_t qdsp_hip_xlate_fir_decim_cf32_out_size(void* h, int64_t count);
int qdsp_hip_xlate_fir_decim_cf32_set_mode(void* h, int mode);
int qdsp_hip_xlate_fir_decim_cf32_reset(void* h);
int qdsp_hip_xlate_fir_decim_cf32_history_len(void* h);
int qdsp_hip_xlate_fir_decim_cf32_get_history(void* h, float* hist_iq);
int qdsp_hip_xlate_fir_decim_cf32_set_history(void* h, const float* hist_iq);
int qdsp_hip_xlate_fir_decim_cf32_history_dev(void* h, void** d_hist);
int qdsp_hip_xlate_fir_decim_cf32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
void qdsp_hip_xlate_fir_decim_cf32_destroy(void* h);

/* ---- channelizer: Splitter -> N x VFO (src/dsp/routing.h:47-57 + src/dsp/vfo.h), batched ---- */
/* N frequency-translating decimators on ONE input stream: channel c is exactly
 * xlate_fir_decim_cf32 with phase increment (phase_inc_re[c], phase_inc_im[c]); all channels
 * share taps / interp / decim.  Output is channel-major: channel c's samples start at
 * out + c*out_stride (complex samples).  process* return the per-channel output count.
 * 64 channels spaced 1/64 turn per sample with decim 64 and <= 256 taps run as ONE polyphase
 * filter-bank kernel (qdsp_amd/csrc/chan.hip); any other plan runs one fused kernel per
 * channel on the same stream.  set_mode(QDSP_HIP_FIR_DIRECT) forces the per-channel form. */
int qdsp_hip_chan_cf32_create(void** h, int device, const float* taps, int ntaps, int interp,
                              int decim, int nchan, const float* phase_inc_re,
                              const float* phase_inc_im, int max_block);
int qdsp_hip_chan_cf32_process(void* h, const float* in_iq, int count, float* out_iq, int out_stride);
int64_t qdsp_hip_chan_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                                       int64_t out_stride, void* hip_stream);
/* The same operator inside a block graph (Splitter -> N x VFO with identical filters, src/dsp/routing.h:47-57): ONE batched
 * launch, channel c's samples into its own stream buffer outs[c] with link code out_links[c] (QDSP_HIP_LINK_HOST_DEFERRED:
 * pinned host buffer, stored by the kernel itself, complete once `done_event` has fired -- or QDSP_HIP_LINK_HOST: complete on
 * return; QDSP_HIP_LINK_DEVICE / _PIPELINED as for *_process_ex).  Always the per-channel (non-uniform) form.  Returns the
 * per-channel output count; QDSP_HIP_ESIZE / _EINVAL when a buffer cannot be served this way (pageable or > 1 MiB host
 * output, taps beyond LDS): the caller then runs the channels one by one. */
int64_t qdsp_hip_chan_cf32_process_links(void* h, const void* in, int in_link, int count, void* const* outs, const int* out_links,
                                         void* done_event);
/* NCO phase (exact) and filter history of channel `chan` into (to_vfo != 0) or out of (to_vfo == 0) a stand-alone
 * xlate_fir_decim_cf32 handle of the same design: how a Splitter hands running VFOs to a bank and back without a glitch.
 * Synchronises the device; histories of different length are zeroed instead of copied. */
int qdsp_hip_chan_cf32_move_channel_state(void* h, int chan, void* vfo_handle, int to_vfo);
int64_t qdsp_hip_chan_cf32_out_size(void* h, int64_t count);
int qdsp_hip_chan_cf32_set_phase_inc(void* h, int chan, float phase_inc_re, float phase_inc_im);
int qdsp_hip_chan_cf32_set_mode(void* h, int mode);
int qdsp_hip_chan_cf32_set_volk_gain(void* h, int on);
int qdsp_hip_chan_cf32_reset(void* h);
/* Time-sharding a stream over GPUs (as for xlate_fir_decim_cf32): history_len raw INPUT
 * samples that precede the next call, rotated internally with each channel's NCO; advance
 * moves every channel's NCO by n samples without processing. */
int qdsp_hip_chan_cf32_history_len(void* h);
int qdsp_hip_chan_cf32_set_history_dev(void* h, const void* d_hist, void* hip_stream);
int qdsp_hip_chan_cf32_advance(void* h, int64_t n);
int qdsp_hip_chan_cf32_channels(void* h);
void qdsp_hip_chan_cf32_destroy(void* h);

/* ---- element-wise two-input blocks: Add / Substract / Multiply (src/dsp/math.h:7-145) ---- */
/* out[i] = a[i] op b[i] over `count` samples; complex_data: samples are interleaved {re, im}
 * pairs and QDSP_HIP_MATH_MUL is the complex product (volk_32fc_x2_multiply_32fc, math.h:127),
 * add / subtract are per float (math.h:33,80).  Results are bit-identical to VOLK's generic
 * kernels (separately rounded products and sums).  Stateless apart from staging buffers.
 * process: host pointers, synchronous; process_ex: each side host or device (1 = device);
 * process_dev: asynchronous on hip_stream, pointers 16-byte aligned. */
#define QDSP_HIP_MATH_ADD 0
#define QDSP_HIP_MATH_SUB 1
#define QDSP_HIP_MATH_MUL 2
int qdsp_hip_math_create(void** h, int device, int op, int complex_data, int max_block);
int qdsp_hip_math_process(void* h, const void* a, const void* b, int count, void* out);
int qdsp_hip_math_process_ex(void* h, const void* a, int a_dev, const void* b, int b_dev, int count,
                             void* out, int out_dev);
int qdsp_hip_math_process_dev(void* h, const void* d_a, const void* d_b, int64_t count, void* d_out,
                              void* hip_stream);
void qdsp_hip_math_destroy(void* h);

/* ---- synthetic IQ source (measurement harness) ------------------------------ */
/* Counter-based uniform [-1,1) per float component, generated on device so benchmarks are
 * HBM->HBM.  Bit-identical to oracle_synth_iq() for the same (first_sample, seed). */
int qdsp_hip_synth_iq_dev(int device, void* d_out_iq, int64_t first_sample, int64_t count,
                          uint32_t seed, void* hip_stream);

/* ---- introspection for the measurement harness ----------------------------------------- */
/* Name of the kernel the last process* call of this handle launched, and its launch
 * geometry; used by bench.py to pick the right row out of a rocprofv3 kernel trace. */
int qdsp_hip_last_kernel(void* h, char* name, int name_len, int* grid, int* block, int* lds_bytes);
/* Time `iters` back-to-back process_dev launches of this handle on `hip_stream` with HIP
 * events recorded on that same stream; returns mean milliseconds per launch in *ms.
 * State advances as for `iters` ordinary calls. */
int qdsp_hip_time_process_dev(void* h, const void* d_in, int64_t count, void* d_out,
                              void* hip_stream, int iters, float* ms);

/* ---- ring: the halo exchange of the time-sharded path over RCCL / xGMI ----------------------------------
 * The reference has no multi-device path (one thread per block: src/dsp/block.h:83-85).  A long stream is cut into time chunks over
 * the GPUs of a node, one process per GPU; a chunk's filter needs the last H INPUT samples of the chunk before it in the stream --
 * the samples src/dsp/filter.h:71 / src/dsp/resampling.h:129 carry from one run() to the next.  Every step each rank posts
 *     ncclGroupStart(); ncclSend(my tail -> rank + 1); ncclRecv(rank - 1's tail); ncclGroupEnd();
 * on the ring's own HIP stream and installs what arrives with <op>_set_history_dev.  A C++ graph (qdsp_amd/host/examples/
 * graph_check.cpp `shard`) and qdsp_amd/sharding.py RingStream call the same five entry points.
 *   available  1 when RCCL could be bound in this process (dlopen of librccl.so.1), else 0: lets every rank vote BEFORE the
 *              collective create (a ring with one end missing would hang; qdsp_amd/sharding.py takes that vote)
 *   unique_id  one rank (rank 0) obtains the 128-byte id; the caller carries it to every rank (file, socket, MPI, a torch store)
 *   create     collective: every rank of the ring calls it with the same id; world 1 = the rank is its own neighbour
 *   post       my tail (halo_bytes at d_tail, as of what `producer_stream` has queued so far) -> rank + 1, and rank - 1's tail ->
 *              the next of four receive buffers; returns at once.  At most two posts may be outstanding.
 *   complete   `consumer_stream` waits for the oldest outstanding post; *d_halo = what arrived with it, *d_prev_halo = what
 *              arrived with the post before it (zeros before the first): in a block-cyclic cut rank 0's predecessor is the LAST
 *              rank of the step before.  Either pointer may be NULL.
 *   drain      host-side wait for everything posted (end of stream: every rank has one exchange in flight that no step reads;
 *              also: call it before a collective of ANOTHER communicator of the same process, e.g. a torch.distributed barrier)
 *   info       what the communicator itself reports: ncclCommCount / ncclCommUserRank / ncclCommCuDevice and the RCCL version
 *              code (-1 where the bound RCCL lacks the query) -- evidence for a scaling run, not needed by the data path
 *   set_timing / exchange_us   optional: HIP timing events around each send/recv group on the ring stream; mean / max duration
 *              in microseconds of the exchanges finished so far (includes waiting for the neighbour to post its end)
 * Lifetimes (one consumer stream per ring):
 *   d_tail        must stay unmodified until the matching complete(): work queued on consumer_stream AFTER that complete may
 *                 overwrite it (the send is ordered before the event complete waits for).
 *   *d_halo       valid for work queued on consumer_stream between this complete() and the next but one complete();
 *   *d_prev_halo  ... between this complete() and the next one.  The ring waits for that work (an event recorded on
 *                 consumer_stream by each complete) before a later post receives into the same buffer.  Work on another
 *                 stream must be ordered behind consumer_stream by the caller.
 * Errors: QDSP_HIP_ERCCL when librccl.so.1 cannot be loaded or an RCCL call fails (its message goes to stderr). */
#define QDSP_HIP_RING_ID_BYTES 128
int qdsp_hip_ring_available(void);
int qdsp_hip_ring_unique_id(void* id);
int qdsp_hip_ring_create(void** ring, int device, int rank, int world, const void* id, int halo_bytes);
int qdsp_hip_ring_post(void* ring, const void* d_tail, void* producer_stream);
int qdsp_hip_ring_complete(void* ring, void* consumer_stream, const void** d_halo, const void** d_prev_halo);
int qdsp_hip_ring_drain(void* ring);
int qdsp_hip_ring_info(void* ring, int* comm_ranks, int* comm_rank, int* comm_device, int* rccl_version);
int qdsp_hip_ring_set_timing(void* ring, int on);
int qdsp_hip_ring_exchange_us(void* ring, double* mean_us, double* max_us, long long* exchanges);
void qdsp_hip_ring_destroy(void* ring);

#ifdef __cplusplus
}
#endif
#endif /* QDSP_HIP_H */
