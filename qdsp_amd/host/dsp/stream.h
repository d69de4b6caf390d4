// dsp/stream.h -- the two-buffer hand-off channel between blocks.
//
// Same public surface and protocol as the reference's dsp::stream<T>
// (src/dsp/stream.h:10-19 untyped_steam, :21-125 stream<T>):
//     producer: fill writeBuf[0..n) ; swap(n)   -- blocks until the consumer flushed the
//               previous block, exchanges the two buffers, wakes the consumer; false = stopping
//     consumer: n = read()                      -- blocks for data; -1 = stopping
//               ... use readBuf[0..n) ... ; flush()
// Exactly one block is in flight per stream; capacity is STREAM_BUFFER_SIZE elements per
// buffer.  What differs from the reference is only where the buffers live: they are pinned
// host memory (qdsp_hip_host_alloc) instead of volk_malloc, so a HIP-backed block can DMA
// straight out of readBuf and into writeBuf; when no HIP runtime is usable they fall back
// to ordinary aligned memory (the channel itself needs no GPU).
//
// Device-resident companion (new): a stream can also carry a PAIR OF DEVICE
// BUFFERS that swap in lockstep with the host pair.  A HIP-backed producer whose consumer is
// another HIP-backed block (`consumerTakesDevice`, set by that consumer when it registers the
// stream) leaves its block in devWriteBuf and marks it `writeOnDevice`; after swap() the
// consumer sees `readOnDevice` and reads devReadBuf: no PCIe round trip between adjacent GPU
// blocks.  Host blocks never look at these members, so the reference protocol is unchanged.
// Between two blocks that both launch before they swap / flush (FIR, PolyphaseResampler, FrequencyXlator, VFO,
// SineSource, Splitter) the link is "pipelined": see the members at the end of the class.  Towards a host
// consumer a HIP-backed producer may swap a block in while its kernel is still running; the completion event it
// recorded travels with the buffer and read() waits for it, so a consumer never sees an unfinished block.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>

#include "qdsp_hip.h"

// Elements per buffer: the reference's value (src/dsp/stream.h:7) unless the build says otherwise.  It is also the most a block hands
// to the library per call, and a GPU call costs 5-10 us whatever its size (EXPERIMENTS.md "Through the block graph"): a graph that is
// free to choose its block size gets more of the kernels' rate with -DSTREAM_BUFFER_SIZE=4194304 or 16777216
// (profiles/r04_graph_bench.txt; qdsp_amd/host/Makefile builds graph_check_big that way).  Every block of a graph must be compiled
// with the same value.
#ifndef STREAM_BUFFER_SIZE
#define STREAM_BUFFER_SIZE 1000000
#endif

namespace dsp {

namespace detail { struct vfo_bank_member; }

// (sic) the reference spells it this way; blocks register streams through this base.
class untyped_steam {
public:
    virtual ~untyped_steam() {}
    virtual bool swap(int size) { (void)size; return false; }
    virtual int read() { return -1; }
    virtual void flush() {}
    virtual void stopWriter() {}
    virtual void clearWriteStop() {}
    virtual void stopReader() {}
    virtual void clearReadStop() {}
};

namespace detail {
struct stream_mem {
    static void* get(size_t bytes, bool& pinned) {
        void* p = nullptr;
        if (qdsp_hip_host_alloc(&p, bytes) == 0 && p) { pinned = true; return p; }
        pinned = false;
        if (posix_memalign(&p, 64, bytes) != 0) { return nullptr; }
        return p;
    }
    static void put(void* p, bool pinned) {
        if (!p) { return; }
        if (pinned) { qdsp_hip_host_free(p); } else { free(p); }
    }
};
}  // namespace detail

template <class T>
class stream : public untyped_steam {
public:
    stream() {
        writeBuf = static_cast<T*>(detail::stream_mem::get(sizeof(T) * STREAM_BUFFER_SIZE, pinnedW));
        readBuf = static_cast<T*>(detail::stream_mem::get(sizeof(T) * STREAM_BUFFER_SIZE, pinnedR));
    }

    ~stream() {
        // the two pointers have been exchanged an unknown number of times; each carries its flag
        detail::stream_mem::put(writeBuf, pinnedW);
        detail::stream_mem::put(readBuf, pinnedR);
        if (devWriteBuf) { qdsp_hip_dev_free(devDevice, devWriteBuf); }
        if (devReadBuf) { qdsp_hip_dev_free(devDevice, devReadBuf); }
    }

    // Allocate the device pair on first use; false if there is no usable device memory.
    bool ensureDevice(int device) {
        if (devWriteBuf && devReadBuf) { return true; }
        void *a = nullptr, *b = nullptr;
        if (qdsp_hip_dev_alloc(device, &a, sizeof(T) * STREAM_BUFFER_SIZE) != 0) { return false; }
        if (qdsp_hip_dev_alloc(device, &b, sizeof(T) * STREAM_BUFFER_SIZE) != 0) { qdsp_hip_dev_free(device, a); return false; }
        devDevice = device;
        devWriteBuf = static_cast<T*>(a);
        devReadBuf = static_cast<T*>(b);
        return true;
    }

    stream(const stream&) = delete;
    stream& operator=(const stream&) = delete;

    bool swap(int size) override {
        std::unique_lock<std::mutex> lk(mtx);
        waitFor(lk, [this] { return slotFree || writerStopped; });
        if (writerStopped) { return false; }
        T* t = writeBuf; writeBuf = readBuf; readBuf = t;
        bool p = pinnedW; pinnedW = pinnedR; pinnedR = p;
        t = devWriteBuf; devWriteBuf = devReadBuf; devReadBuf = t;
        readOnDevice = writeOnDevice;
        writeOnDevice = false;
        readPipelined = writePipelined;
        writePipelined = false;
        readDoneEvt = writeDoneEvt;
        writeDoneEvt = nullptr;
        readIsToken = writeIsToken;
        writeIsToken = false;
        pending = size;
        slotFree = false;
        hasData = true;
        lk.unlock();
        cv.notify_all();
        return true;
    }

    int read() override {
        std::unique_lock<std::mutex> lk(mtx);
        waitFor(lk, [this] { return hasData || readerStopped; });
        if (readerStopped) { return -1; }
        const int n = pending;
        void* evt = readDoneEvt;
        readDoneEvt = nullptr;
        lk.unlock();
        // a GPU producer may have handed the block over with its kernel still running (QDSP_HIP_LINK_HOST_DEFERRED):
        // the buffer is the consumer's only once that event has fired
        if (evt) { (void)qdsp_hip_event_wait(evt); }
        return n;
    }

    void flush() override {
        {
            std::lock_guard<std::mutex> lk(mtx);
            hasData = false;
            slotFree = true;
        }
        cv.notify_all();
    }

    void stopWriter() override { setFlag(writerStopped, true); }
    void clearWriteStop() override { setFlag(writerStopped, false); }
    void stopReader() override { setFlag(readerStopped, true); }
    void clearReadStop() override { setFlag(readerStopped, false); }

    T* writeBuf;
    T* readBuf;

    // device-resident companion (see the header comment)
    T* devWriteBuf = nullptr;
    T* devReadBuf = nullptr;
    bool writeOnDevice = false;        // producer: the block being swapped in lives in devWriteBuf
    bool readOnDevice = false;         // consumer: the block just read lives in devReadBuf
    // set by a HIP-backed consumer on its input stream, cleared when it lets go of the stream or a host consumer
    // attaches (claimConsumer / releaseConsumer below); read by the producer's worker thread while setInput() on a
    // control thread may change it, hence atomic
    std::atomic<bool> consumerTakesDevice{false};
    // Pipelined link (QDSP_HIP_LINK_PIPELINED, qdsp_hip.h): producer and consumer both launch into the library's
    // in-order stream before they swap / flush, so the producer does not wait for its kernel and the GPU runs the
    // two back to back; the host threads only exchange buffers.
    bool writePipelined = false;       // producer: the block being swapped in may still be in flight on that stream
    bool readPipelined = false;        // consumer: ... so read it on the same stream
    std::atomic<bool> consumerPipelined{false};    // set by a consumer that launches into that stream before it flushes
    // The consumer end of the link.  A HIP-backed block claims the stream when it registers it as its input
    // (takesDevice: it reads devReadBuf; pipelined: it launches before it flushes); a host consumer (sinks, any
    // reference block) and a consumer that re-plumbs or dies while live release it, so the producer goes back to
    // filling the host buffers.
    void claimConsumer(bool takesDevice, bool pipelined) {
        consumerTakesDevice.store(takesDevice);
        consumerPipelined.store(takesDevice && pipelined);
    }
    void releaseConsumer() { claimConsumer(false, false); }
    // Splitter -> N x VFO banking (routing.h, vfo.h): a VFO core hangs its descriptor on its input stream; a Splitter
    // that finds identical VFOs behind all its outputs runs them as ONE batched launch and only sends token blocks down
    // these links (vfo_bank.h).  Set before the graph starts (VFO::init), read by the Splitter's worker.
    std::shared_ptr<detail::vfo_bank_member> bankMember;
    // Block until the consumer has released the block it was given (flush()), false if the writer side is being stopped.
    bool waitFlushed() {
        std::unique_lock<std::mutex> lk(mtx);
        waitFor(lk, [this] { return slotFree || writerStopped; });
        return !writerStopped;
    }
    int linkIn() const { return readOnDevice ? (readPipelined ? QDSP_HIP_LINK_PIPELINED : QDSP_HIP_LINK_DEVICE) : QDSP_HIP_LINK_HOST; }
    int linkOut(bool outDev) const { return outDev ? (consumerPipelined ? QDSP_HIP_LINK_PIPELINED : QDSP_HIP_LINK_DEVICE) : QDSP_HIP_LINK_HOST; }
    void markWritten(int link, void* doneEvt = nullptr) {
        writeOnDevice = link == QDSP_HIP_LINK_DEVICE || link == QDSP_HIP_LINK_PIPELINED;
        writePipelined = link == QDSP_HIP_LINK_PIPELINED;
        writeDoneEvt = link == QDSP_HIP_LINK_HOST_DEFERRED ? doneEvt : nullptr;
    }
    // Deferred completion of a host block: set by the producer, waited for inside read()
    void* writeDoneEvt = nullptr;
    void* readDoneEvt = nullptr;
    // A block that carries a count and no samples (Splitter -> banked VFO, vfo_bank.h).  The mark travels WITH the block, like the
    // link codes above: a consumer recognises a token whatever has happened to the bank between the producer's swap() and its
    // own read() (the Splitter may have been re-plumbed or destroyed in between -- ADVICE round 2).
    bool writeIsToken = false;
    bool readIsToken = false;
    void markToken() { writeIsToken = true; }

private:
    // A GPU-backed neighbour answers within tens of microseconds, less than a futex sleep and wake-up costs:
    // poll for a bounded time (QDSP_STREAM_SPIN_US, default 60) before blocking on the condition variable.  Same protocol, same
    // wake-up conditions as the reference's cv.wait; an idle graph still sleeps.
    static int spinMicros() {
        static const int v = [] { const char* e = getenv("QDSP_STREAM_SPIN_US"); return e ? atoi(e) : 60; }();
        return v;
    }
    template <class PRED> void waitFor(std::unique_lock<std::mutex>& lk, PRED pred) {
        if (pred()) { return; }
        const int kSpinMicros = spinMicros();
        if (kSpinMicros <= 0) { cv.wait(lk, pred); return; }
        const auto t0 = std::chrono::steady_clock::now();
        do {
            lk.unlock();
            for (int i = 0; i < 64; i++) { __asm__ __volatile__("" ::: "memory"); }
            lk.lock();
            if (pred()) { return; }
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(kSpinMicros));
        cv.wait(lk, pred);
    }

    void setFlag(bool& f, bool v) {
        {
            std::lock_guard<std::mutex> lk(mtx);
            f = v;
        }
        cv.notify_all();
    }

    std::mutex mtx;
    std::condition_variable cv;
    bool slotFree = true;      // consumer has released the read buffer (reference: canSwap)
    bool hasData = false;      // a swapped block is waiting to be read (reference: dataReady)
    bool writerStopped = false;
    bool readerStopped = false;
    int pending = 0;
    bool pinnedW = false, pinnedR = false;
    int devDevice = 0;
};

}  // namespace dsp
