// chan_ops.hip -- the channelizer operator (Splitter -> N x VFO in the reference: src/dsp/routing.h:47-57 + src/dsp/vfo.h:19-36):
// the uniform 64-channel polyphase plan (chan.hip), the batched per-channel kernels for any other plan, and qdsp_hip_chan_cf32_*.
// Split out of qdsp_hip.hip in round 3 (engine.hip.h).
#include "engine.hip.h"

namespace qh {

// ---- channelizer: N fused VFOs on one input (Splitter -> N x VFO in the reference) ----------
void chan_destroy(Chan* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (Engine* e : c->vfo) destroy(e);
    if (c->d_taps) (void)hipFree(c->d_taps);
    if (c->d_tw64) (void)hipFree(c->d_tw64);
    if (c->d_phases) (void)hipFree(c->d_phases);
    if (c->d_batch) (void)hipFree(c->d_batch);
    if (c->d_batch_mf) (void)hipFree(c->d_batch_mf);
    for (int i = 0; i < 2; i++)
        if (c->d_hist[i]) (void)hipFree(c->d_hist[i]);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->magic = 0;
    delete c;
}
// The uniform plan the fast path needs: 64 channels, interp 1, decim 64, <= 256 taps, and the
// channels' fixed-point phase increments equal to channel 0's plus c * (+-2^58) to within
// float rounding of the (cos, sin) pairs they came from.  Fills sign / deviations.
bool chan_uniform_plan(const Chan* c, int* inv, long long* ddelta) {
    if (c->nchan != 64 || c->interp != 1 || c->ntaps < 1 || c->ntaps > 256) return false;
    if (c->decim != 64 && c->decim != 32 && c->decim != 16 && c->decim != 8) return false;
    const unsigned long long d0 = c->vfo[0]->dphase;
    const long long tol = (long long)(18446744073709551616.0 * 4e-7);
    for (int sign = 1; sign >= -1; sign -= 2) {
        bool ok = true;
        for (int i = 0; i < 64 && ok; i++) {
            const unsigned long long ideal = d0 + (unsigned long long)((long long)sign * (long long)i) * (1ULL << 58);
            const long long dev = (long long)(c->vfo[i]->dphase - ideal);
            if (dev > tol || dev < -tol) ok = false;
            ddelta[i] = dev;
        }
        if (ok) { *inv = sign > 0; return true; }
    }
    return false;
}

// Tables and the shared history of the uniform fast path (allocated on first use).
int chan_uniform_prepare(Chan* c) {
    const int P = c->ntaps;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    if (!c->d_taps) {
        std::vector<float2> tw(64);
        for (int m = 0; m < 64; m++) tw[m] = make_float2((float)cosl(two_pi * m / 64), (float)(-sinl(two_pi * m / 64)));
        HIPCHK(hipMalloc(&c->d_taps, 256 * sizeof(float2)));
        HIPCHK(hipMalloc(&c->d_tw64, 64 * sizeof(float2)));
        HIPCHK(hipMemcpy(c->d_tw64, tw.data(), 64 * sizeof(float2), hipMemcpyHostToDevice));
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipMalloc(&c->d_hist[i], (size_t)P * sizeof(float2)));
            HIPCHK(hipMemset(c->d_hist[i], 0, (size_t)P * sizeof(float2)));
        }
    }
    const unsigned long long d0 = c->vfo[0]->dphase;
    if (!c->gt_valid || c->gt_dphase != d0) {
        // channel 0's mixer folded into the prototype: g[k] = h[k] * exp(j 2pi k dphase_0 / 2^64)
        std::vector<float2> g(256, make_float2(0.0f, 0.0f));
        for (int k = 0; k < c->ntaps; k++) {
            double cr, sr;
            unit_of_fx(d0, (long double)k, &cr, &sr);
            const double h = c->vfo[0]->taps_host[k];
            g[k] = make_float2((float)(h * cr), (float)(h * sr));
        }
        HIPCHK(hipDeviceSynchronize());     // rare (retune of channel 0): nothing in flight may still read the table
        HIPCHK(hipMemcpy(c->d_taps, g.data(), 256 * sizeof(float2), hipMemcpyHostToDevice));
        c->gt_dphase = d0;
        c->gt_valid = true;
    }
    return 0;
}

int chan_launch_uniform(Chan* c, const void* d_in, int64_t count, int64_t nout, void* d_out, int64_t out_stride, hipStream_t s) {
    qk::ChanArgs a;
    memset(&a, 0, sizeof(a));
    if (!chan_uniform_plan(c, &a.inv, a.ddelta)) return QDSP_HIP_EINVAL;
    const int P = c->ntaps;
    { int rc = chan_uniform_prepare(c); if (rc) return rc; }
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(c->d_hist[c->cur]);
    a.hist_next = reinterpret_cast<float2*>(c->d_hist[c->cur ^ 1]);
    a.gtaps = c->d_taps;
    a.tw64 = c->d_tw64;
    a.count = count;
    a.nout = nout;
    a.out_stride = out_stride;
    a.P = P;
    a.Q = (c->ntaps + 63) / 64;
    a.M = c->decim;
    a.ntiles = (int)((nout + 15) / 16);
    a.st4 = ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0 && (out_stride & 1) == 0 && !qk::knob(qk::K_CHAN_NO_ST4, 0)) ? 1 : 0;
    a.waves = 4;
    int nwg = 256 * 48;  // 3 resident per CU, 16 rounds (round 3: 12 -> 48: -2 %, profiles/r03_chan_tuning.txt; 3 / 6 / 12 / 48 re-measured in round 4: 2.556 / 2.507 / 2.489 / 2.440 ms)
    if (nwg > (a.ntiles + a.waves - 1) / a.waves) nwg = (a.ntiles + a.waves - 1) / a.waves;
    if (nwg < 1) nwg = 1;
    a.nwg = nwg;
    a.kcentre = (c->ntaps - 1) / 2;
    a.phase0 = c->vfo[0]->phase;
    a.dphase0 = c->vfo[0]->dphase;
    for (int i = 0; i < 64; i++) {
        a.dphi[i] = c->vfo[i]->phase - c->vfo[0]->phase;
        a.gm1[i] = c->volk_gain ? c->vfo[i]->gm1 : 0.0f;
    }
    {   // second-order term of the per-output deviation rotation: only when 15 output times turn a channel by more than 1e-4 rad
        long long dmax = 0;
        for (int i = 0; i < 64; i++) dmax = std::max(dmax, a.ddelta[i] < 0 ? -a.ddelta[i] : a.ddelta[i]);
        a.quad = 15.0 * (double)dmax * (double)a.M * 3.4061215800865545e-19 > 1e-4 || 0;
    }
    a.abl = qk::knob(qk::K_CHAN_ABL, 0);
    const size_t lds = qk::chan_uniform_lds_bytes(a.waves);
    int rc = qk::launch_chan_uniform(a, nwg + 1, s);
    if (rc) return rc;
    c->cur ^= 1;
    for (int i = 0; i < 64; i++) c->vfo[i]->phase += (unsigned long long)count * c->vfo[i]->dphase;
    c->last.name = "chan_uniform_kernel";
    c->last.grid = nwg + 1;
    c->last.block = 64 * a.waves;
    c->last.lds = (int)lds;
    return 0;
}

// Non-uniform plans whose design the MFMA decimator serves (large integer decimations: the VFO bank's usual shape):
// ALL channels in one launch of decim_mfma_batch_kernel.  Returns 1 if it does not apply.
int chan_launch_batch_mf(Chan* c, const void* d_in, int64_t count, int64_t nout, void* d_out, int64_t out_stride, hipStream_t s,
                         void* const* out_ptrs) {
    Engine* e0 = c->vfo[0];
    if (!e0->d_taps_mf || nout <= 0 || qk::knob(qk::K_NO_MF, 0) || qk::knob(qk::K_NO_MF_BATCH, 0)) return 1;
    // per wave: taps, slot table and an FP64 sincos before the first tile -- small launches are quicker on the general
    // direct kernel (profiles/r02_tune_chan_batch.txt: even at 4 channels x 1e6 samples, 16 x 1e6: 36 against 55 us)
    if (count * c->nchan < (int64_t)qk::knob(qk::K_MF_BATCH_MIN_WORK, 1 << 22)) return 1;
    for (Engine* e : c->vfo)
        if (e->cur != e0->cur || !e->rotate || e->ch != 2 || !e->d_taps_mf || e->mf_KJ != e0->mf_KJ || e->mf_QS != 1 || e->mf_keep2) return 1;
    std::vector<qk::MfChanConst> key((size_t)c->nchan);
    for (int i = 0; i < c->nchan; i++) {
        Engine* e = c->vfo[i];
        qk::MfChanConst& k = key[i];
        memset(&k, 0, sizeof(k));
        k.hist[0] = reinterpret_cast<const float2*>(e->d_hist[0]);
        k.hist[1] = reinterpret_cast<const float2*>(e->d_hist[1]);
        k.dphase = e->dphase;
        k.gm1 = e->volk_gain ? e->gm1 : 0.0f;
    }
    bool dirty = !c->d_batch_mf || c->batch_mf_key.size() != key.size();
    for (size_t i = 0; !dirty && i < key.size(); i++)
        dirty = key[i].hist[0] != c->batch_mf_key[i].hist[0] || key[i].hist[1] != c->batch_mf_key[i].hist[1] ||
                key[i].dphase != c->batch_mf_key[i].dphase || key[i].gm1 != c->batch_mf_key[i].gm1;
    if (dirty) {
        for (size_t i = 0; i < key.size(); i++) mf_rot_tables(key[i].dphase, e0->M, e0->mf_KJ, &key[i].rot_step, key[i].rot_k);
        HIPCHK(hipDeviceSynchronize());     // (rare -- a retune: nothing in flight may still read the old table)
        if (c->d_batch_mf && c->batch_mf_key.size() != key.size()) { HIPCHK(hipFree(c->d_batch_mf)); c->d_batch_mf = nullptr; }
        if (!c->d_batch_mf) HIPCHK(hipMalloc(&c->d_batch_mf, key.size() * sizeof(qk::MfChanConst)));
        HIPCHK(hipMemcpy(c->d_batch_mf, key.data(), key.size() * sizeof(qk::MfChanConst), hipMemcpyHostToDevice));
        c->batch_mf_key = key;
    }
    qk::MfBatchArgs b;
    memset(&b, 0, sizeof(b));
    qk::MfArgs& a = b.a;
    a.in = static_cast<const float2*>(d_in);
    a.tapk = e0->d_taps_mf;
    a.count = count;
    a.nout = nout;
    a.P = e0->P;
    a.M = e0->M;
    mf_tasks(a, nout, c->nchan, true);
    b.out_stride = out_stride;
    b.cur = e0->cur;
    const int depth = qk::knob(qk::K_MF_DEPTH, e0->mf_KJ <= 8 ? 2 : 1);
    for (int base = 0; base < c->nchan; base += qk::kMfBatchMax) {
        const int nb = (c->nchan - base < qk::kMfBatchMax) ? c->nchan - base : qk::kMfBatchMax;
        b.tab = c->d_batch_mf + base;
        a.out = out_ptrs ? nullptr : static_cast<float2*>(d_out) + (size_t)base * out_stride;
        b.use_ptrs = out_ptrs ? 1 : 0;
        for (int i = 0; i < nb; i++) {
            b.phase0[i] = c->vfo[base + i]->phase;
            b.outs[i] = out_ptrs ? out_ptrs[base + i] : nullptr;
        }
        const int rc = qk::launch_mf_dec_batch(b, nb, e0->mf_KJ, depth, s);
        if (rc) return rc < 0 && rc != -1 ? rc : QDSP_HIP_EINVAL;
    }
    for (Engine* e : c->vfo) {
        e->cur ^= 1;
        e->phase += (unsigned long long)count * e->dphase;
        e->raw_valid = false;
    }
    c->last.name = "decim_mfma_batch_kernel";
    c->last.grid = ((a.ntasks + 3) / 4 + 1) * c->nchan;
    c->last.block = 256;
    c->last.lds = 4 * (16 * (8 * e0->mf_KJ + 2) + 64) * (int)sizeof(float2);
    return 0;
}

// Non-uniform plans (arbitrary offsets, any channel count): ALL channels in one launch of resamp_any_batch_kernel
// (blockIdx.y = channel).  Returns 1 if the batched form does not apply (the caller then loops over the channels).
int chan_launch_batch(Chan* c, const void* d_in, int64_t count, int64_t nout, void* d_out, int64_t out_stride, hipStream_t s,
                      void* const* out_ptrs = nullptr) {
    {
        const int rc = chan_launch_batch_mf(c, d_in, count, nout, d_out, out_stride, s, out_ptrs);
        if (rc <= 0) return rc;
    }
    constexpr int NT = 256;
    Engine* e0 = c->vfo[0];
    for (Engine* e : c->vfo)
        if (e->cur != e0->cur || !e->rotate || e->ch != 2) return 1;
    qk::AnyBatchArgs b;
    memset(&b, 0, sizeof(b));
    qk::AnyArgs& a = b.a;
    a.L = e0->L;
    a.M = e0->M;
    a.P = e0->P;
    a.count = count;
    a.nout = nout;
    bool lt = false, pad = false;
    const size_t lds = fill_any_geometry(a, 2, &lt, &pad, (int)c->vfo.size());
    if (lds == 0) return 1;
    if (!c->d_phases) {
        // the reference's [interp][tapsPerPhase] table (buildTapPhases, resampling.h:137-166), whatever layout the
        // per-channel engines keep their taps in
        const int L = e0->L, P = e0->P, ntaps = e0->ntaps;
        std::vector<float> host((size_t)L * P, 0.0f);
        int curt = 0;
        for (int tap = 0; tap < P; tap++)
            for (int phase = 0; phase < L; phase++)
                host[(size_t)((L - 1) - phase) * P + tap] = (curt < ntaps) ? e0->taps_host[curt++] : 0.0f;
        HIPCHK(hipMalloc(&c->d_phases, host.size() * sizeof(float)));
        HIPCHK(hipMemcpy(c->d_phases, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // per-channel constants: rebuilt and uploaded only when something they were built from changed
    std::vector<qk::AnyChanConst> key((size_t)c->nchan);
    for (int i = 0; i < c->nchan; i++) {
        Engine* e = c->vfo[i];
        qk::AnyChanConst& k = key[i];
        memset(&k, 0, sizeof(k));
        k.hist[0] = e->d_hist[0];
        k.hist[1] = e->d_hist[1];
        k.dphase = e->dphase;
        k.gm1 = e->volk_gain ? e->gm1 : 0.0f;
    }
    bool dirty = !c->d_batch || c->batch_key.size() != key.size();
    for (size_t i = 0; !dirty && i < key.size(); i++)
        dirty = key[i].hist[0] != c->batch_key[i].hist[0] || key[i].hist[1] != c->batch_key[i].hist[1] ||
                key[i].dphase != c->batch_key[i].dphase || key[i].gm1 != c->batch_key[i].gm1;
    if (dirty) {
        for (size_t i = 0; i < key.size(); i++) {
            qk::AnyArgs tmp;
            memset(&tmp, 0, sizeof(tmp));
            tmp.dphase = key[i].dphase;
            fill_stage_rot(tmp, NT);
            key[i].rot_nt = tmp.rot_nt;
            key[i].rot_8nt = tmp.rot_8nt;
            for (int k = 0; k < 8; k++) key[i].rot_k[k] = tmp.rot_k[k];
        }
        HIPCHK(hipDeviceSynchronize());     // (rare -- a retune: nothing in flight may still read the old table)
        if (c->d_batch && c->batch_key.size() != key.size()) { HIPCHK(hipFree(c->d_batch)); c->d_batch = nullptr; }
        if (!c->d_batch) HIPCHK(hipMalloc(&c->d_batch, key.size() * sizeof(qk::AnyChanConst)));
        HIPCHK(hipMemcpy(c->d_batch, key.data(), key.size() * sizeof(qk::AnyChanConst), hipMemcpyHostToDevice));
        c->batch_key = key;
    }
    a.in = d_in;
    a.phases = c->d_phases;
    // persistent workgroups per channel: the grid's x extent times the channels should fill the chip a few times over
    int nwg = (256 * 8 + c->nchan - 1) / c->nchan;
    if (nwg > a.nblocks) nwg = a.nblocks;
    if (nwg < 1) nwg = 1;
    a.nwg = nwg;
    b.out_stride = out_stride;
    b.cur = e0->cur;
    for (int base = 0; base < c->nchan; base += qk::kAnyBatchMax) {
        const int nb = (c->nchan - base < qk::kAnyBatchMax) ? c->nchan - base : qk::kAnyBatchMax;
        b.tab = c->d_batch + base;
        a.out = out_ptrs ? nullptr : static_cast<void*>(static_cast<float2*>(d_out) + (size_t)base * out_stride);
        b.use_ptrs = out_ptrs ? 1 : 0;
        for (int i = 0; i < nb; i++) {
            b.phase0[i] = c->vfo[base + i]->phase;
            b.outs[i] = out_ptrs ? out_ptrs[base + i] : nullptr;
        }
        const dim3 grid(nwg + 1, nb);
        if (pad) {
            if (lt) hipLaunchKernelGGL((qk::resamp_any_batch_kernel<NT, true, true>), grid, dim3(NT), lds, s, b);
            else hipLaunchKernelGGL((qk::resamp_any_batch_kernel<NT, false, true>), grid, dim3(NT), lds, s, b);
        } else {
            if (lt) hipLaunchKernelGGL((qk::resamp_any_batch_kernel<NT, true, false>), grid, dim3(NT), lds, s, b);
            else hipLaunchKernelGGL((qk::resamp_any_batch_kernel<NT, false, false>), grid, dim3(NT), lds, s, b);
        }
        HIPCHK(hipGetLastError());
    }
    for (Engine* e : c->vfo) {
        e->cur ^= 1;
        e->phase += (unsigned long long)count * e->dphase;
        e->raw_valid = false;
    }
    c->last.name = "resamp_any_batch_kernel";
    c->last.grid = (nwg + 1) * c->nchan;
    c->last.block = NT;
    c->last.lds = (int)lds;
    return 0;
}

// Batch or loop?  One launch wins wherever a per-channel call is launch- or latency-bound (reference-sized blocks:
// 16 channels of a 1e6-sample block 403 us as 16 launches) and wherever resamp_any_kernel is what a channel would run
// anyway; chip-filling calls of plans that have a faster dedicated kernel (overlap-save, strided-window) keep it.
bool chan_batch_wins(const Chan* c, int64_t count) {
    if (c->nchan < 2 || qk::knob(qk::K_NO_CHAN_BATCH, 0)) return false;
    const Engine* e = c->vfo[0];
    if (count <= (int64_t)(1 << 22)) return true;
    if (e->d_taps_mf && !qk::knob(qk::K_NO_MF, 0) && !qk::knob(qk::K_NO_MF_BATCH, 0)) return true;   // what each channel would run anyway
    const bool dedicated = (fft_eligible(e, count) || use_win(e) || use_core(e) || use_lm(e));
    return !dedicated;
}

int64_t chan_process_dev(Chan* c, const void* d_in, int64_t count, void* d_out, int64_t out_stride, void* stream) {
    if (count < 0 || c->vfo.empty()) return QDSP_HIP_EINVAL;
    for (Engine* e : c->vfo) apply_pending_inc(e);
    const int64_t nout = out_size(c->vfo[0], count);
    if (out_stride < nout) return QDSP_HIP_EINVAL;
    {
        int inv;
        long long dd[64];
        const int mode = c->mode ? c->mode : qk::knob(qk::K_FIR_MODE, 0);
        if (mode != 1 && chan_uniform_plan(c, &inv, dd)) {
            int rc = chan_launch_uniform(c, d_in, count, nout, d_out, out_stride, static_cast<hipStream_t>(stream));
            return rc ? rc : nout;
        }
    }
    {
        const int mode = c->mode ? c->mode : qk::knob(qk::K_FIR_MODE, 0);
        if (mode == 0 && count > 0 && chan_batch_wins(c, count)) {
            const int rc = chan_launch_batch(c, d_in, count, nout, d_out, out_stride, static_cast<hipStream_t>(stream));
            if (rc == 0) return nout;
            if (rc < 0) return rc;
        }
    }
    for (int i = 0; i < c->nchan; i++) {
        float2* o = static_cast<float2*>(d_out) + (size_t)i * out_stride;
        const int64_t r = process_dev(c->vfo[i], d_in, count, o, stream);
        if (r < 0) return r;
    }
    c->last = c->vfo[0]->last;
    return nout;
}

}  // namespace qh

using namespace qh;

extern "C" {

// ---- channelizer ------------------------------------------------------------------------------
int qdsp_hip_chan_cf32_create(void** h, int device, const float* taps, int ntaps, int interp, int decim, int nchan,
                              const float* phase_inc_re, const float* phase_inc_im, int max_block) {
    if (!h || nchan <= 0 || !phase_inc_re || !phase_inc_im) return QDSP_HIP_EINVAL;
    *h = nullptr;
    Chan* c = new (std::nothrow) Chan();
    if (!c) return QDSP_HIP_ENOMEM;
    c->device = device;
    c->nchan = nchan;
    c->ntaps = ntaps;
    c->interp = interp;
    c->decim = decim;
    int rc = 0;
    for (int i = 0; i < nchan && rc == 0; i++) {
        void* eh = nullptr;
        rc = qdsp_hip_xlate_fir_decim_cf32_create(&eh, device, taps, ntaps, interp, decim, phase_inc_re[i], phase_inc_im[i], 0);
        if (rc == 0) c->vfo.push_back(static_cast<Engine*>(eh));
    }
    if (rc == 0 && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) rc = QDSP_HIP_ENOMEM;
    if (rc == 0 && max_block > 0) {
        const size_t oc = (size_t)out_size(c->vfo[0], max_block) + 1;
        if (hipMalloc(&c->d_in, (size_t)max_block * 8) != hipSuccess || hipMalloc(&c->d_out, oc * 8 * nchan) != hipSuccess) rc = QDSP_HIP_ENOMEM;
        c->max_block = max_block;
        c->d_in_cap = (size_t)max_block;
        c->out_cap = oc;
    }
    if (rc) { chan_destroy(c); return rc; }
    *h = c;
    return 0;
}
int64_t qdsp_hip_chan_cf32_out_size(void* h, int64_t count) {
    Chan* c = as_chan(h);
    return c ? out_size(c->vfo[0], count) : QDSP_HIP_EINVAL;
}
int64_t qdsp_hip_chan_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out, int64_t out_stride, void* s) {
    Chan* c = as_chan(h);
    if (!c) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    return chan_process_dev(c, d_in, count, d_out, out_stride, s);
}
int qdsp_hip_chan_cf32_process(void* h, const float* in, int count, float* out, int out_stride) {
    Chan* c = as_chan(h);
    if (!c || count < 0 || count > c->max_block) return c ? QDSP_HIP_ESIZE : QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    if (count) HIPCHK(hipMemcpyAsync(c->d_in, in, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
    const int64_t nout = chan_process_dev(c, c->d_in, count, c->d_out, (int64_t)c->out_cap, c->stream);
    if (nout < 0) return (int)nout;
    if (out_stride < nout) return QDSP_HIP_EINVAL;
    if (nout)
        HIPCHK(hipMemcpy2DAsync(out, (size_t)out_stride * 8, c->d_out, c->out_cap * 8, (size_t)nout * 8, c->nchan,
                                hipMemcpyDeviceToHost, c->stream));
    HIPCHK(wait_stream(c->stream));
    return (int)nout;
}
// Splitter -> N x VFO inside a block graph: one batched launch, every channel's output into its OWN stream buffer.
int64_t qdsp_hip_chan_cf32_process_links(void* h, const void* in, int in_link, int count, void* const* outs, const int* out_links,
                                         void* done_event) {
    Chan* c = as_chan(h);
    if (!c || count < 0 || !outs || !out_links || (count > 0 && !in)) return QDSP_HIP_EINVAL;
    if (c->vfo.empty()) return QDSP_HIP_EINVAL;
    for (Engine* e : c->vfo) apply_pending_inc(e);
    HIPCHK(hipSetDevice(c->device));
    const int64_t nout = out_size(c->vfo[0], count);
    if (count == 0) return 0;
    hipStream_t st = c->stream;
    bool any_pipe = in_link == QDSP_HIP_LINK_PIPELINED, any_host = false, any_dev_sync = false;
    for (int i = 0; i < c->nchan; i++) {
        if (out_links[i] == QDSP_HIP_LINK_PIPELINED) any_pipe = true;
        else if (out_links[i] == QDSP_HIP_LINK_DEVICE) any_dev_sync = true;
        else if (out_links[i] == QDSP_HIP_LINK_HOST_DEFERRED || out_links[i] == QDSP_HIP_LINK_HOST) any_host = true;
        else return QDSP_HIP_EINVAL;
    }
    if (any_pipe) {
        st = shared_stream(c->device);
        if (!st) return QDSP_HIP_ENOMEM;
    }
    // host outputs are stored by the kernel itself into the pinned, device-mapped stream buffers (as process_ex does
    // for results up to QDSP_HIP_DIRECT_OUT_MAX_BYTES); anything else is not served here -- the caller falls back
    std::vector<void*> dst((size_t)c->nchan);
    const size_t out_bytes = (size_t)nout * sizeof(float2);
    for (int i = 0; i < c->nchan; i++) {
        if (out_links[i] == QDSP_HIP_LINK_HOST_DEFERRED || out_links[i] == QDSP_HIP_LINK_HOST) {
            if (out_bytes > (size_t)(1 << 20)) return QDSP_HIP_ESIZE;
            void* m = out_bytes ? mapped_host_ptr(outs[i]) : outs[i];
            if (!m) return QDSP_HIP_ESIZE;
            dst[i] = m;
        } else {
            dst[i] = outs[i];
        }
    }
    const void* src = in;
    if (in_link == QDSP_HIP_LINK_HOST) {
        // the staging buffer of THIS entry point has its own capacity: max_block stays what d_out (sized at creation) can hold,
        // so a later qdsp_hip_chan_cf32_process(count <= max_block) never writes past d_out
        if ((size_t)count > c->d_in_cap || !c->d_in) {
            if (c->d_in) HIPCHK(hipFree(c->d_in));
            c->d_in = nullptr;
            c->d_in_cap = 0;
            HIPCHK(hipMalloc(&c->d_in, (size_t)count * sizeof(float2)));
            c->d_in_cap = (size_t)count;
        }
        HIPCHK(hipMemcpyAsync(c->d_in, in, (size_t)count * sizeof(float2), hipMemcpyHostToDevice, st));
        src = c->d_in;
    }
    const int rc = chan_launch_batch(c, src, count, nout, nullptr, 0, st, dst.data());
    if (rc != 0) return rc < 0 ? rc : QDSP_HIP_EINVAL;
    bool must_wait = in_link != QDSP_HIP_LINK_PIPELINED || any_dev_sync;   // a host / plain device input is released on return
    if (any_host) {
        if (done_event) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(done_event), st));
        else must_wait = true;
        for (int i = 0; i < c->nchan && !must_wait; i++)
            if (out_links[i] == QDSP_HIP_LINK_HOST) must_wait = true;       // not deferred: complete on return
    }
    if (must_wait) HIPCHK(hipStreamSynchronize(st));
    return nout;
}
// State hand-over between channel `chan` of a bank and a stand-alone fused-VFO handle of the same design: the NCO
// phase (exact, 64-bit fixed point) and the filter history.  to_vfo != 0: channel -> handle, else handle -> channel.
int qdsp_hip_chan_cf32_move_channel_state(void* h, int chan, void* vfo, int to_vfo) {
    Chan* c = as_chan(h);
    Engine* v = as_engine(vfo, KIND_VFO);
    if (!c || !v || chan < 0 || chan >= c->nchan) return QDSP_HIP_EINVAL;
    Engine* e = c->vfo[chan];
    if (e->device != v->device) return QDSP_HIP_EINVAL;
    apply_pending_inc(e);
    apply_pending_inc(v);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());          // (rare: a bank is built or taken down)
    Engine* src = to_vfo ? e : v;
    Engine* dst = to_vfo ? v : e;
    dst->phase = src->phase;
    dst->raw_valid = false;
    if (src->H == dst->H && src->H > 0 && src->ch == dst->ch)
        HIPCHK(hipMemcpy(dst->d_hist[dst->cur], src->d_hist[src->cur], (size_t)src->H * src->ch * sizeof(float), hipMemcpyDeviceToDevice));
    else if (dst->H > 0)
        HIPCHK(hipMemset(dst->d_hist[dst->cur], 0, (size_t)dst->H * dst->ch * sizeof(float)));
    return 0;
}
int qdsp_hip_chan_cf32_set_phase_inc(void* h, int chan, float re, float im) {
    Chan* c = as_chan(h);
    if (!c || chan < 0 || chan >= c->nchan || (re == 0.0f && im == 0.0f)) return QDSP_HIP_EINVAL;
    set_inc(c->vfo[chan], re, im);
    return 0;
}
int qdsp_hip_chan_cf32_set_mode(void* h, int mode) {
    Chan* c = as_chan(h);
    if (!c || mode < 0 || mode > 2) return QDSP_HIP_EINVAL;
    c->mode = mode;
    for (Engine* e : c->vfo) e->fir_mode = mode;
    return 0;
}
int qdsp_hip_chan_cf32_set_volk_gain(void* h, int on) {
    Chan* c = as_chan(h);
    if (!c) return QDSP_HIP_EINVAL;
    c->volk_gain = on != 0;
    for (Engine* e : c->vfo) e->volk_gain = on != 0;
    return 0;
}
int qdsp_hip_chan_cf32_reset(void* h) {
    Chan* c = as_chan(h);
    if (!c) return QDSP_HIP_EINVAL;
    for (Engine* e : c->vfo) { int rc = reset(e); if (rc) return rc; }
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < 2; i++)
        if (c->d_hist[i]) HIPCHK(hipMemset(c->d_hist[i], 0, (size_t)c->ntaps * sizeof(float2)));
    return 0;
}
int qdsp_hip_chan_cf32_history_len(void* h) {
    Chan* c = as_chan(h);
    return c ? c->vfo[0]->H : QDSP_HIP_EINVAL;
}
int qdsp_hip_chan_cf32_set_history_dev(void* h, const void* d_hist, void* s) {
    Chan* c = as_chan(h);
    if (!c || !d_hist) return QDSP_HIP_EINVAL;
    for (Engine* e : c->vfo) apply_pending_inc(e);
    HIPCHK(hipSetDevice(c->device));
    int inv;
    long long dd[64];
    if (chan_uniform_plan(c, &inv, dd)) {
        // shared history of the fast path: raw input (channel 0's mixer lives in the taps)
        int rc = chan_uniform_prepare(c);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->d_hist[c->cur], d_hist, (size_t)c->ntaps * sizeof(float2), hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(s)));
    }
    const int mode = c->mode ? c->mode : qk::knob(qk::K_FIR_MODE, 0);
    if (mode == 1 || !chan_uniform_plan(c, &inv, dd))
        for (Engine* e : c->vfo) { int rc = set_history_dev(e, d_hist, s); if (rc) return rc; }
    return 0;
}
int qdsp_hip_chan_cf32_advance(void* h, int64_t n) {
    Chan* c = as_chan(h);
    if (!c) return QDSP_HIP_EINVAL;
    for (Engine* e : c->vfo) { apply_pending_inc(e); e->phase += (unsigned long long)n * e->dphase; e->raw_valid = false; }
    return 0;
}
int qdsp_hip_chan_cf32_channels(void* h) {
    Chan* c = as_chan(h);
    return c ? c->nchan : QDSP_HIP_EINVAL;
}
void qdsp_hip_chan_cf32_destroy(void* h) { Chan* c = as_chan(h); if (c) chan_destroy(c); }

}  // extern "C"
