// dsp/processing.h -- dsp::FrequencyXlator<T>, HIP-backed (the reference's other blocks in
// this header -- AGC, squelch, packer ... -- are serial recurrences outside the hot path and
// are not provided).
//
// Drop-in for src/dsp/processing.h:10-81.  init()/setSampleRate()/setFrequency() compute
// phaseDelta exactly as the reference does -- theta = (freq/sampleRate) * 2.0f * FL_M_PI in
// float, then the float cos/sin (processing.h:20,39,48) -- and hand the pair to the GPU
// NCO (qdsp_hip_xlate_cf32_*), which advances by arg(phaseDelta) per sample with a 64-bit
// fixed-point accumulator instead of VOLK's recursive float phasor.
#pragma once
#include <cmath>
#include <type_traits>

#include "block.h"
#include "filter.h"

namespace dsp {

template <class T>
class FrequencyXlator : public generic_block<FrequencyXlator<T>> {
    static_assert(std::is_same<T, complex_t>::value, "FrequencyXlator is implemented for complex_t (as in the reference)");
    using base = generic_block<FrequencyXlator<T>>;

public:
    FrequencyXlator() {}
    FrequencyXlator(stream<complex_t>* in, float sampleRate, float freq) { init(in, sampleRate, freq); }

    ~FrequencyXlator() {
        const bool live = base::running;
        base::stop();
        if (live && _in) { _in->releaseConsumer(); }
        if (handle) { qdsp_hip_xlate_cf32_destroy(handle); }
    }

    void init(stream<complex_t>* in, float sampleRate, float freq) {
        _in = in;
        _sampleRate = sampleRate;
        _freq = freq;
        computeDelta();
        const int rc = qdsp_hip_xlate_cf32_create(&handle, detail::hipDeviceForBlocks(), deltaRe, deltaIm, STREAM_BUFFER_SIZE);
        if (rc != 0) { handle = nullptr; detail::hipBlockFail("FrequencyXlator::init", rc); }
        base::registerInput(_in);
        base::registerOutput(&out);
        _in->claimConsumer(handle != nullptr, true);
    }

    // (sic) the reference names its input setter setInputSize (processing.h:26)
    void setInputSize(stream<complex_t>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in->releaseConsumer();
        _in = in;
        _in->claimConsumer(handle != nullptr, true);
        base::registerInput(_in);
        base::tempStart();
    }

    void setSampleRate(float sampleRate) {
        _sampleRate = sampleRate;
        pushDelta();
    }
    float getSampleRate() { return _sampleRate; }

    void setFrequency(float freq) {
        _freq = freq;
        pushDelta();
    }
    float getFrequency() { return _freq; }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        if (!handle) { return -1; }
        const bool inDev = _in->readOnDevice;
        const bool outDev = out.consumerTakesDevice && out.ensureDevice(detail::hipDeviceForBlocks());
        const void* src = inDev ? static_cast<const void*>(_in->devReadBuf) : static_cast<const void*>(_in->readBuf);
        void* dst = outDev ? static_cast<void*>(out.devWriteBuf) : static_cast<void*>(out.writeBuf);
        void* evt = nullptr;
        const int outLink = outDev ? out.linkOut(true) : done.arm(handle, evt);
        const int rc = qdsp_hip_xlate_cf32_process_ex(handle, src, _in->linkIn(), count, dst, outLink);
        _in->flush();
        if (rc != 0) { return detail::hipBlockFail("FrequencyXlator::run", rc); }
        out.markWritten(outLink, evt);
        if (!out.swap(count)) { return -1; }
        return count;
    }

    stream<complex_t> out;

private:
    void computeDelta() {
        const float theta = (_freq / _sampleRate) * 2.0f * FL_M_PI;
        deltaRe = std::cos(theta);
        deltaIm = std::sin(theta);
    }
    // The reference changes phaseDelta while the worker runs ("No need to restart"); the
    // device NCO takes the new increment between two run() calls the same way.
    void pushDelta() {
        computeDelta();
        if (handle) { qdsp_hip_xlate_cf32_set_phase_inc(handle, deltaRe, deltaIm); }
    }

    float _sampleRate = 1.0f, _freq = 0.0f;
    float deltaRe = 1.0f, deltaIm = 0.0f;
    stream<complex_t>* _in = nullptr;
    void* handle = nullptr;
    detail::done_events done;
};

}  // namespace dsp
