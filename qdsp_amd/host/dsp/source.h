// dsp/source.h -- SineSource (HIP-backed) and HandlerSource<T>.
//
// SineSource: same surface as the reference (src/dsp/source.h:5-71); where the reference
// rotates a buffer of ones with VOLK each run() (source.h:55-59), this one asks the device
// NCO for the next _blockSize samples (qdsp_hip_sine_cf32_*), device-resident when the
// consumer is another HIP-backed block.
// HandlerSource<T>: a callback fills out.writeBuf and returns the count (source.h:74-107).
#pragma once
#include <cmath>

#include "block.h"
#include "filter.h"

namespace dsp {

class SineSource : public generic_block<SineSource> {
    using base = generic_block<SineSource>;

public:
    SineSource() {}
    SineSource(int blockSize, float sampleRate, float freq) { init(blockSize, sampleRate, freq); }

    ~SineSource() {
        base::stop();
        if (handle) { qdsp_hip_sine_cf32_destroy(handle); }
    }

    void init(int blockSize, float sampleRate, float freq) {
        _blockSize = blockSize;
        _sampleRate = sampleRate;
        _freq = freq;
        float dRe, dIm;
        delta(dRe, dIm);
        const int rc = qdsp_hip_sine_cf32_create(&handle, detail::hipDeviceForBlocks(), dRe, dIm, STREAM_BUFFER_SIZE);
        if (rc != 0) { handle = nullptr; detail::hipBlockFail("SineSource::init", rc); }
        base::registerOutput(&out);
    }

    void setBlockSize(int blockSize) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _blockSize = blockSize;
        base::tempStart();
    }
    int getBlockSize() { return _blockSize; }

    // (both retune the running NCO between two run() calls; the phase carries on)
    void setSampleRate(float sampleRate) { retune(sampleRate, _freq); }
    void setFrequency(float freq) { retune(_sampleRate, freq); }
    float getSampleRate() { return _sampleRate; }
    float getFrequency() { return _freq; }

    int run() override {
        if (!handle) { return -1; }
        const bool outDev = out.consumerTakesDevice && out.ensureDevice(detail::hipDeviceForBlocks());
        void* dst = outDev ? static_cast<void*>(out.devWriteBuf) : static_cast<void*>(out.writeBuf);
        void* evt = nullptr;
        const int outLink = outDev ? out.linkOut(true) : done.arm(handle, evt);
        const int rc = qdsp_hip_sine_cf32_generate(handle, _blockSize, dst, outLink);
        if (rc != 0) { return detail::hipBlockFail("SineSource::run", rc); }
        out.markWritten(outLink, evt);
        if (!out.swap(_blockSize)) { return -1; }
        return _blockSize;
    }

    stream<complex_t> out;

private:
    void delta(float& dRe, float& dIm) const {
        const float theta = (_freq / _sampleRate) * 2.0f * FL_M_PI;  // source.h:18
        dRe = std::cos(theta);
        dIm = std::sin(theta);
    }
    void retune(float sampleRate, float freq) {
        _sampleRate = sampleRate;
        _freq = freq;
        float dRe, dIm;
        delta(dRe, dIm);
        if (handle) { qdsp_hip_sine_cf32_set_phase_inc(handle, dRe, dIm); }
    }

    int _blockSize = 0;
    float _sampleRate = 1.0f, _freq = 0.0f;
    void* handle = nullptr;
    detail::done_events done;
};

template <class T>
class HandlerSource : public generic_block<HandlerSource<T>> {
    using base = generic_block<HandlerSource<T>>;

public:
    // fills `data` (room for STREAM_BUFFER_SIZE samples) and returns how many it wrote, < 0 to end the stream
    using fill_fn = int (*)(T* data, void* ctx);

    HandlerSource() {}
    HandlerSource(fill_fn handler, void* ctx) { init(handler, ctx); }

    void init(fill_fn handler, void* ctx) {
        bind(handler, ctx);
        base::registerOutput(&out);
    }

    void setHandler(fill_fn handler, void* ctx) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        bind(handler, ctx);
        base::tempStart();
    }

    int run() override {
        const int count = _handler(out.writeBuf, _ctx);
        if (count < 0) { return -1; }
        // The reference ignores swap()'s result here (source.h:97-101), so its worker spins
        // forever once stop() has set the writer-stop flag and stop() never joins; ending the
        // loop on a refused swap is the one deliberate deviation in this file.
        if (!out.swap(count)) { return -1; }
        return count;
    }

    stream<T> out;

private:
    void bind(fill_fn handler, void* ctx) {
        _handler = handler;
        _ctx = ctx;
    }

    fill_fn _handler = nullptr;
    void* _ctx = nullptr;
};

}  // namespace dsp
