// dsp/resampling.h -- dsp::PolyphaseResampler<T>, HIP-backed.
//
// Drop-in for src/dsp/resampling.h:9-189: rational interp/decim resampler (a decimator when
// interp == 1).  Rate handling follows the reference: interp = outSR/gcd, decim = inSR/gcd
// on the int-cast rates (resampling.h:28-30), taps designed by the window with gain
// `interp` (resampling.h:34), phase split and per-block phase restart inside the library
// (qdsp_hip_decim_*).  T = complex_t, stereo_t (same kernels: a float pair with real taps)
// or float.
#pragma once
#include <numeric>
#include <type_traits>
#include <vector>

#include "block.h"
#include "filter.h"
#include "window.h"

namespace dsp {

template <class T>
class PolyphaseResampler : public generic_block<PolyphaseResampler<T>> {
    using base = generic_block<PolyphaseResampler<T>>;
    static constexpr bool kPair = std::is_same<T, complex_t>::value || std::is_same<T, stereo_t>::value;
    static_assert(kPair || std::is_same<T, float>::value, "PolyphaseResampler<T>: T is complex_t, stereo_t or float");

public:
    PolyphaseResampler() {}
    PolyphaseResampler(stream<T>* in, dsp::filter_window::generic_window* window, float inSampleRate, float outSampleRate) {
        init(in, window, inSampleRate, outSampleRate);
    }

    ~PolyphaseResampler() {
        const bool live = base::running;
        base::stop();
        if (live && _in) { _in->releaseConsumer(); }
        if (handle) { kPair ? qdsp_hip_decim_cf32_destroy(handle) : qdsp_hip_decim_f32_destroy(handle); }
    }

    void init(stream<T>* in, dsp::filter_window::generic_window* window, float inSampleRate, float outSampleRate) {
        _in = in;
        _window = window;
        _inSampleRate = inSampleRate;
        _outSampleRate = outSampleRate;
        updateRatio();
        designTaps();
        const int dev = detail::hipDeviceForBlocks();
        const int rc = kPair ? qdsp_hip_decim_cf32_create(&handle, dev, taps.data(), (int)taps.size(), _interp, _decim, STREAM_BUFFER_SIZE)
                             : qdsp_hip_decim_f32_create(&handle, dev, taps.data(), (int)taps.size(), _interp, _decim, STREAM_BUFFER_SIZE);
        if (rc != 0) { handle = nullptr; detail::hipBlockFail("PolyphaseResampler::init", rc); }
        base::registerInput(_in);
        base::registerOutput(&out);
        _in->claimConsumer(handle != nullptr, true);
    }

    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in->releaseConsumer();
        _in = in;
        _in->claimConsumer(handle != nullptr, true);
        base::registerInput(_in);
        base::tempStart();
    }

    // The reference re-splits the EXISTING taps for the new ratio without redesigning them
    // (resampling.h:53-73); so does this.
    void setInSampleRate(float inSampleRate) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _inSampleRate = inSampleRate;
        updateRatio();
        reconfigure();
        base::tempStart();
    }

    void setOutSampleRate(float outSampleRate) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _outSampleRate = outSampleRate;
        updateRatio();
        reconfigure();
        base::tempStart();
    }

    int getInterpolation() { return _interp; }
    int getDecimation() { return _decim; }

    void updateWindow(dsp::filter_window::generic_window* window) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _window = window;
        designTaps();
        reconfigure();
        base::tempStart();
    }

    int calcOutSize(int in) override { return (in * _interp) / _decim; }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        if (!handle) { return -1; }
        const bool inDev = _in->readOnDevice;
        const bool outDev = out.consumerTakesDevice && out.ensureDevice(detail::hipDeviceForBlocks());
        const void* src = inDev ? static_cast<const void*>(_in->devReadBuf) : static_cast<const void*>(_in->readBuf);
        void* dst = outDev ? static_cast<void*>(out.devWriteBuf) : static_cast<void*>(out.writeBuf);
        void* evt = nullptr;
        const int inLink = _in->linkIn(), outLink = outDev ? out.linkOut(true) : done.arm(handle, evt);
        const int outCount = kPair ? qdsp_hip_decim_cf32_process_ex(handle, src, inLink, count, dst, outLink)
                                   : qdsp_hip_decim_f32_process_ex(handle, src, inLink, count, dst, outLink);
        _in->flush();
        if (outCount < 0) { return detail::hipBlockFail("PolyphaseResampler::run", outCount); }
        out.markWritten(outLink, evt);
        if (!out.swap(outCount)) { return -1; }
        return count;
    }

    stream<T> out;

private:
    void updateRatio() {
        const int g = std::gcd((int)_inSampleRate, (int)_outSampleRate);
        _interp = _outSampleRate / g;
        _decim = _inSampleRate / g;
    }

    void designTaps() {
        const int n = _window->getTapCount();
        taps.assign(n > 0 ? (size_t)n + 1 : 1, 0.0f);
        _window->createTaps(taps.data(), n, _interp);
        taps.resize(n > 0 ? n : 0);
    }

    void reconfigure() {
        if (!handle) { return; }
        const int rc = kPair ? qdsp_hip_decim_cf32_configure(handle, taps.data(), (int)taps.size(), _interp, _decim)
                             : qdsp_hip_decim_f32_configure(handle, taps.data(), (int)taps.size(), _interp, _decim);
        if (rc != 0) { detail::hipBlockFail("PolyphaseResampler::reconfigure", rc); }
    }

    stream<T>* _in = nullptr;
    dsp::filter_window::generic_window* _window = nullptr;
    int _interp = 1, _decim = 1;
    float _inSampleRate = 1.0f, _outSampleRate = 1.0f;
    std::vector<float> taps;
    void* handle = nullptr;
    detail::done_events done;
};

}  // namespace dsp
