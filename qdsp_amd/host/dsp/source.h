// dsp/source.h -- HandlerSource<T>: a callback fills out.writeBuf and returns the count
// (reference: src/dsp/source.h:74-107).  (SineSource is VOLK-rotator based and listed as
// "next" in SURVEY 8f; it is not part of this round.)
#pragma once
#include "block.h"

namespace dsp {

template <class T>
class HandlerSource : public generic_block<HandlerSource<T>> {
    using base = generic_block<HandlerSource<T>>;

public:
    HandlerSource() {}
    HandlerSource(int (*handler)(T* data, void* ctx), void* ctx) { init(handler, ctx); }

    void init(int (*handler)(T* data, void* ctx), void* ctx) {
        _handler = handler;
        _ctx = ctx;
        base::registerOutput(&out);
    }

    void setHandler(int (*handler)(T* data, void* ctx), void* ctx) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _handler = handler;
        _ctx = ctx;
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
    int (*_handler)(T* data, void* ctx) = nullptr;
    void* _ctx = nullptr;
};

}  // namespace dsp
