// dsp/sink.h -- HandlerSink<T>, NullSink<T>, FileSink<T>: blocks with one input and no output
// (reference: src/dsp/sink.h:7-49, :96-132, :134-180).  The three differ only in what they do
// with a block once it has been read, so the read/flush loop and the input re-plumbing live
// in one CRTP base and each sink supplies consume().
#pragma once
#include <fstream>
#include <string>

#include "block.h"

namespace dsp {

namespace detail {
template <class T, class SINK>
class sink_base : public generic_block<sink_base<T, SINK>> {
protected:
    // (the block runtime befriends its BLOCK parameter: that is this base, which does the plumbing)
    using base = generic_block<sink_base<T, SINK>>;

    void attach(stream<T>* in) {
        _in = in;
        _in->releaseConsumer();   // a host consumer: whatever GPU block read this stream before, the producer fills readBuf again
        base::registerInput(_in);
    }

public:
    // swap the upstream stream while the graph is live (worker paused around the change)
    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> guard(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        attach(in);
        base::tempStart();
    }

    int run() override {
        const int n = _in->read();
        if (n < 0) { return -1; }
        static_cast<SINK*>(this)->consume(_in->readBuf, n);
        _in->flush();
        return n;
    }

protected:
    stream<T>* _in = nullptr;
};
}  // namespace detail

// Hands every block to a user callback.
template <class T>
class HandlerSink : public detail::sink_base<T, HandlerSink<T>> {
    using sb = detail::sink_base<T, HandlerSink<T>>;
    friend sb;

public:
    using handler_t = void (*)(T* data, int count, void* ctx);

    HandlerSink() {}
    HandlerSink(stream<T>* in, handler_t handler, void* ctx) { init(in, handler, ctx); }

    void init(stream<T>* in, handler_t handler, void* ctx) {
        _handler = handler;
        _ctx = ctx;
        sb::attach(in);
    }

    void setHandler(handler_t handler, void* ctx) {
        std::lock_guard<std::mutex> guard(sb::base::ctrlMtx);
        sb::base::tempStop();
        _handler = handler;
        _ctx = ctx;
        sb::base::tempStart();
    }

private:
    void consume(T* data, int n) { _handler(data, n, _ctx); }

    handler_t _handler = nullptr;
    void* _ctx = nullptr;
};

// Discards everything (keeps an upstream block running).
template <class T>
class NullSink : public detail::sink_base<T, NullSink<T>> {
    using sb = detail::sink_base<T, NullSink<T>>;
    friend sb;

public:
    NullSink() {}
    NullSink(stream<T>* in) { init(in); }
    void init(stream<T>* in) { sb::attach(in); }

private:
    void consume(T*, int) {}
};

// Appends the raw samples to a binary file.
template <class T>
class FileSink : public detail::sink_base<T, FileSink<T>> {
    using sb = detail::sink_base<T, FileSink<T>>;
    friend sb;

public:
    FileSink() {}
    FileSink(stream<T>* in, std::string path) { init(in, path); }
    ~FileSink() {
        sb::base::stop();
        if (file.is_open()) { file.close(); }
    }

    void init(stream<T>* in, std::string path) {
        file.open(path, std::ios::binary);
        sb::attach(in);
    }

    bool isOpen() { return file.is_open(); }

private:
    void consume(T* data, int n) {
        if (file.is_open()) { file.write(reinterpret_cast<const char*>(data), (std::streamsize)n * sizeof(T)); }
    }

    std::ofstream file;
};

}  // namespace dsp
