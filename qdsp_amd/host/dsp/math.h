// dsp/math.h -- dsp::Add<T>, dsp::Substract<T> (sic), dsp::Multiply<T>, HIP-backed.
//
// Drop-in for src/dsp/math.h:7-145: two input streams, one output; a run() reads one block from
// each input and, as the reference does, DROPS the pair (flush both, return 0) when their sizes
// differ (math.h:26-30).  Arithmetic: qdsp_hip_math_* -- add / subtract per float (complex and
// stereo samples are two floats, math.h:33,80), Multiply<complex_t> the complex product
// (math.h:127), Multiply<float> per element.  Multiply<stereo_t> does not instantiate in the
// reference either (it would hand stereo_t* to volk_32f_x2_multiply_32f, math.h:130).
// Either input may arrive device-resident (stream.h companion buffers); the output stays on the
// device when its consumer takes device blocks.
#pragma once
#include <type_traits>

#include "block.h"
#include "filter.h"

namespace dsp {

namespace detail {

// (the worker-loop base is instantiated on math_block itself: generic_block befriends exactly its
// template argument, and it is this class that registers the streams)
template <class T, int OP>
class math_block : public generic_block<math_block<T, OP>> {
    static_assert(std::is_same<T, float>::value || std::is_same<T, complex_t>::value || std::is_same<T, stereo_t>::value,
                  "math blocks take float, complex_t or stereo_t streams");
    static_assert(OP != QDSP_HIP_MATH_MUL || !std::is_same<T, stereo_t>::value, "Multiply<stereo_t> is ill-formed in the reference too");
    using base = generic_block<math_block<T, OP>>;

public:
    math_block() {}
    ~math_block() {
        const bool live = base::running;
        base::stop();
        if (live) {
            if (_a) { _a->releaseConsumer(); }
            if (_b) { _b->releaseConsumer(); }
        }
        if (handle) { qdsp_hip_math_destroy(handle); }
    }

    void init(stream<T>* a, stream<T>* b) {
        _a = a;
        _b = b;
        const int rc = qdsp_hip_math_create(&handle, hipDeviceForBlocks(), OP, std::is_same<T, float>::value ? 0 : 1, STREAM_BUFFER_SIZE);
        if (rc != 0) { handle = nullptr; hipBlockFail("math block init", rc); }
        base::registerInput(a);
        base::registerInput(b);
        base::registerOutput(&out);
        _a->claimConsumer(handle != nullptr, false);
        _b->claimConsumer(handle != nullptr, false);
    }

    int run() override {
        const int a_count = _a->read();
        if (a_count < 0) { return -1; }
        const int b_count = _b->read();
        if (b_count < 0) { return -1; }
        if (a_count != b_count) {
            _a->flush();
            _b->flush();
            return 0;
        }
        if (!handle) { return -1; }
        const bool aDev = _a->readOnDevice, bDev = _b->readOnDevice;
        const bool outDev = out.consumerTakesDevice && out.ensureDevice(hipDeviceForBlocks());
        const void* pa = aDev ? static_cast<const void*>(_a->devReadBuf) : static_cast<const void*>(_a->readBuf);
        const void* pb = bDev ? static_cast<const void*>(_b->devReadBuf) : static_cast<const void*>(_b->readBuf);
        void* dst = outDev ? static_cast<void*>(out.devWriteBuf) : static_cast<void*>(out.writeBuf);
        const int rc = qdsp_hip_math_process_ex(handle, pa, aDev, pb, bDev, a_count, dst, outDev);
        _a->flush();
        _b->flush();
        if (rc != 0) { return hipBlockFail("math block run", rc); }
        out.writeOnDevice = outDev;
        if (!out.swap(a_count)) { return -1; }
        return a_count;
    }

    stream<T> out;

private:
    stream<T>* _a = nullptr;
    stream<T>* _b = nullptr;
    void* handle = nullptr;
};

}  // namespace detail

template <class T>
class Add : public detail::math_block<T, QDSP_HIP_MATH_ADD> {
public:
    Add() {}
    Add(stream<T>* a, stream<T>* b) { this->init(a, b); }
};

// (sic) the reference spells it Substract (math.h:54)
template <class T>
class Substract : public detail::math_block<T, QDSP_HIP_MATH_SUB> {
public:
    Substract() {}
    Substract(stream<T>* a, stream<T>* b) { this->init(a, b); }
};

template <class T>
class Multiply : public detail::math_block<T, QDSP_HIP_MATH_MUL> {
public:
    Multiply() {}
    Multiply(stream<T>* a, stream<T>* b) { this->init(a, b); }
};

}  // namespace dsp
