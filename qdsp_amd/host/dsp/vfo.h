// dsp/vfo.h -- dsp::VFO: frequency-translate by -offset, low-pass, resample.
//
// Same public surface as the reference (src/dsp/vfo.h:15-101): init(), start()/stop(), the
// rate / offset / bandwidth setters and `stream<complex_t>* out`.  The reference builds it
// from two blocks on two threads, FrequencyXlator -> (stream hop) -> PolyphaseResampler
// (vfo.h:28-35); here ONE block runs ONE fused kernel (qdsp_hip_xlate_fir_decim_cf32_*):
// the rotation is applied while the tile is staged into LDS, then the polyphase dot
// products run.  Filter design is the reference's: BlackmanWindow with cutoff =
// transition = min(bandWidth, inSR, outSR)/2 at sample rate inSR*interp (vfo.h:26-33).
//
// (The reference never sets its `running` flag, so its VFO::stop() is a no-op and its
// setters never pause the blocks; this one tracks the flag properly.)
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "block.h"
#include "filter.h"
#include "vfo_bank.h"
#include "window.h"

namespace dsp {

namespace detail {
// The fused worker: reads IQ, writes translated + resampled IQ.
class XlatingResampler : public generic_block<XlatingResampler> {
    using base = generic_block<XlatingResampler>;

public:
    XlatingResampler() {}
    ~XlatingResampler() {
        const bool live = base::running;
        base::stop();
        {
            // under designMtx: the Splitter's dropBank / buildBank test `alive` and use `handle` / `out` under the same mutex, so they
            // never hand a destroyed engine to qdsp_hip_chan_cf32_move_channel_state
            std::lock_guard<std::mutex> lk(member->designMtx);
            member->alive.store(false);
            member->handle = nullptr;
            member->out = nullptr;
        }
        if (auto ctl = std::atomic_load(&member->ctl)) { ctl->broken.store(true); }
        if (live && _in) { _in->releaseConsumer(); }
        if (handle) { qdsp_hip_xlate_fir_decim_cf32_destroy(handle); }
    }

    void init(stream<complex_t>* in, const std::vector<float>& taps, int interp, int decim, float dRe, float dIm) {
        _in = in;
        const int rc = qdsp_hip_xlate_fir_decim_cf32_create(&handle, hipDeviceForBlocks(), taps.data(), (int)taps.size(), interp,
                                                            decim, dRe, dIm, STREAM_BUFFER_SIZE);
        if (rc != 0) { handle = nullptr; hipBlockFail("VFO::init", rc); }
        {
            std::lock_guard<std::mutex> lk(member->designMtx);
            member->taps = taps;
            member->interp = interp;
            member->decim = decim;
            member->dRe = dRe;
            member->dIm = dIm;
            member->out = &out;
            member->handle = handle;
        }
        if (handle) { _in->bankMember = member; }     // (a Splitter upstream may run identical VFOs as one batched launch: vfo_bank.h)
        base::registerInput(_in);
        base::registerOutput(&out);
        _in->claimConsumer(handle != nullptr, true);
    }

    void configure(const std::vector<float>& taps, int interp, int decim) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        {
            std::lock_guard<std::mutex> lk(member->designMtx);
            member->taps = taps;
            member->interp = interp;
            member->decim = decim;
        }
        if (auto ctl = std::atomic_load(&member->ctl)) { ctl->broken.store(true); }   // a bank built for the old design is taken down by its Splitter
        if (handle) {
            const int rc = qdsp_hip_xlate_fir_decim_cf32_configure(handle, taps.data(), (int)taps.size(), interp, decim);
            if (rc != 0) { hipBlockFail("VFO::configure", rc); }
        }
        base::tempStart();
    }

    void setPhaseInc(float dRe, float dIm) {
        if (handle) { qdsp_hip_xlate_fir_decim_cf32_set_phase_inc(handle, dRe, dIm); }
        {
            std::lock_guard<std::mutex> lk(member->designMtx);
            member->dRe = dRe;
            member->dIm = dIm;
        }
        if (auto ctl = std::atomic_load(&member->ctl)) {
            std::lock_guard<std::mutex> lk(ctl->m);
            if (ctl->bank) { qdsp_hip_chan_cf32_set_phase_inc(ctl->bank, member->index, dRe, dIm); }
        }
    }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        if (!handle) { return -1; }
        if (_in->readIsToken) {
            // banked (Splitter -> N x identical VFO): the block is a token, the Splitter's batched launch has already put
            // this channel's samples into out's write buffer.  Flush AFTER the swap: the Splitter takes the flush as
            // "this member's write buffer is free again".  The test is the mark on the BLOCK, not the state of the bank: a
            // Splitter that is re-plumbed (bindStream / unbindStream) or destroyed takes its bank down -- moving the channel
            // state, already past this block, back into `handle` -- while this token may still be waiting in the link; running
            // the core's own kernel on it would filter a never-written buffer and advance the NCO a second time.
            out.markWritten(member->outLink, member->evt);
            const bool ok = out.swap(member->outCount);
            _in->flush();
            return ok ? count : -1;
        }
        const bool inDev = _in->readOnDevice;
        const bool outDev = out.consumerTakesDevice && out.ensureDevice(hipDeviceForBlocks());
        const void* src = inDev ? static_cast<const void*>(_in->devReadBuf) : static_cast<const void*>(_in->readBuf);
        void* dst = outDev ? static_cast<void*>(out.devWriteBuf) : static_cast<void*>(out.writeBuf);
        void* evt = nullptr;
        const int outLink = outDev ? out.linkOut(true) : done.arm(handle, evt);
        const int outCount = qdsp_hip_xlate_fir_decim_cf32_process_ex(handle, src, _in->linkIn(), count, dst, outLink);
        _in->flush();
        if (outCount < 0) { return hipBlockFail("VFO::run", outCount); }
        out.markWritten(outLink, evt);
        if (!out.swap(outCount)) { return -1; }
        return count;
    }

    stream<complex_t> out;
    std::shared_ptr<vfo_bank_member> member = std::make_shared<vfo_bank_member>();

private:
    stream<complex_t>* _in = nullptr;
    void* handle = nullptr;
    detail::done_events done;
};
}  // namespace detail

class VFO {
public:
    VFO() {}
    ~VFO() { stop(); }

    VFO(stream<complex_t>* in, float offset, float inSampleRate, float outSampleRate, float bandWidth) {
        init(in, offset, inSampleRate, outSampleRate, bandWidth);
    }

    void init(stream<complex_t>* in, float offset, float inSampleRate, float outSampleRate, float bandWidth) {
        _in = in;
        _offset = offset;
        _inSampleRate = inSampleRate;
        _outSampleRate = outSampleRate;
        _bandWidth = bandWidth;
        // the reference's first design pass (win at inSR) only serves to size things; the
        // taps that run come from the second one at inSR*interp (vfo.h:29-33)
        redesign();
        float dRe, dIm;
        delta(dRe, dIm);
        core.init(_in, taps, _interp, _decim, dRe, dIm);
        out = &core.out;
    }

    void start() {
        if (running) { return; }
        core.start();
        running = true;
    }

    void stop() {
        if (!running) { return; }
        core.stop();
        running = false;
    }

    void setInSampleRate(float inSampleRate) {
        _inSampleRate = inSampleRate;
        redesign();
        float dRe, dIm;
        delta(dRe, dIm);
        core.setPhaseInc(dRe, dIm);
        core.configure(taps, _interp, _decim);
    }

    void setOutSampleRate(float outSampleRate) {
        _outSampleRate = outSampleRate;
        redesign();
        core.configure(taps, _interp, _decim);
    }

    void setOutSampleRate(float outSampleRate, float bandWidth) {
        _outSampleRate = outSampleRate;
        _bandWidth = bandWidth;
        redesign();
        core.configure(taps, _interp, _decim);
    }

    void setOffset(float offset) {
        _offset = offset;
        float dRe, dIm;
        delta(dRe, dIm);
        core.setPhaseInc(dRe, dIm);
    }

    void setBandwidth(float bandWidth) {
        _bandWidth = bandWidth;
        redesign();
        core.configure(taps, _interp, _decim);
    }

    int getInterpolation() const { return _interp; }
    int getDecimation() const { return _decim; }
    const std::vector<float>& getTaps() const { return taps; }

    stream<complex_t>* out = nullptr;

private:
    void redesign() {
        const int g = std::gcd((int)_inSampleRate, (int)_outSampleRate);  // resampling.h:28-30
        _interp = _outSampleRate / g;
        _decim = _inSampleRate / g;
        const float realCutoff = std::min<float>(_bandWidth, std::min<float>(_inSampleRate, _outSampleRate)) / 2.0f;
        win.init(realCutoff, realCutoff, _inSampleRate * _interp);
        const int n = win.getTapCount();
        taps.assign((size_t)n, 0.0f);
        win.createTaps(taps.data(), n, _interp);
    }

    void delta(float& dRe, float& dIm) const {
        const float theta = (-_offset / _inSampleRate) * 2.0f * FL_M_PI;  // xlator.init(_in, inSR, -offset), vfo.h:28
        dRe = std::cos(theta);
        dIm = std::sin(theta);
    }

    bool running = false;
    float _offset = 0, _inSampleRate = 1, _outSampleRate = 1, _bandWidth = 1;
    int _interp = 1, _decim = 1;
    filter_window::BlackmanWindow win;
    std::vector<float> taps;
    stream<complex_t>* _in = nullptr;
    detail::XlatingResampler core;
};

}  // namespace dsp
