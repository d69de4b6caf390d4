// dsp/sink.h -- HandlerSink<T>, NullSink<T>, FileSink<T> (reference: src/dsp/sink.h:7-49,
// :96-132, :134-180).
#pragma once
#include <fstream>
#include <string>

#include "block.h"

namespace dsp {

template <class T>
class HandlerSink : public generic_block<HandlerSink<T>> {
    using base = generic_block<HandlerSink<T>>;

public:
    HandlerSink() {}
    HandlerSink(stream<T>* in, void (*handler)(T* data, int count, void* ctx), void* ctx) { init(in, handler, ctx); }

    void init(stream<T>* in, void (*handler)(T* data, int count, void* ctx), void* ctx) {
        _in = in;
        _handler = handler;
        _ctx = ctx;
        base::registerInput(_in);
    }

    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in = in;
        base::registerInput(_in);
        base::tempStart();
    }

    void setHandler(void (*handler)(T* data, int count, void* ctx), void* ctx) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        _handler = handler;
        _ctx = ctx;
        base::tempStart();
    }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        _handler(_in->readBuf, count, _ctx);
        _in->flush();
        return count;
    }

private:
    stream<T>* _in = nullptr;
    void (*_handler)(T* data, int count, void* ctx) = nullptr;
    void* _ctx = nullptr;
};

template <class T>
class NullSink : public generic_block<NullSink<T>> {
    using base = generic_block<NullSink<T>>;

public:
    NullSink() {}
    NullSink(stream<T>* in) { init(in); }

    void init(stream<T>* in) {
        _in = in;
        base::registerInput(_in);
    }

    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in = in;
        base::registerInput(_in);
        base::tempStart();
    }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        _in->flush();
        return count;
    }

private:
    stream<T>* _in = nullptr;
};

template <class T>
class FileSink : public generic_block<FileSink<T>> {
    using base = generic_block<FileSink<T>>;

public:
    FileSink() {}
    FileSink(stream<T>* in, std::string path) { init(in, path); }
    ~FileSink() {
        base::stop();
        if (file.is_open()) { file.close(); }
    }

    void init(stream<T>* in, std::string path) {
        _in = in;
        file = std::ofstream(path, std::ios::binary);
        base::registerInput(_in);
    }

    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in = in;
        base::registerInput(_in);
        base::tempStart();
    }

    bool isOpen() { return file.is_open(); }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        if (file.is_open()) { file.write(reinterpret_cast<const char*>(_in->readBuf), (std::streamsize)count * sizeof(T)); }
        _in->flush();
        return count;
    }

private:
    stream<T>* _in = nullptr;
    std::ofstream file;
};

}  // namespace dsp
