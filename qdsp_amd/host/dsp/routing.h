// dsp/routing.h -- Splitter<T>: fan one stream out to any number of bound streams
// (reference: src/dsp/routing.h:10-62; its Reshaper is outside the hot path and not provided).
//
// Same surface: init()/setInput()/bindStream()/unbindStream(), private run().  The reference
// memcpy's the block into every bound stream's writeBuf (routing.h:51-54).  Here each bound
// stream gets the block where its consumer wants it: a HIP-backed consumer
// (`consumerTakesDevice`) receives a device-to-device copy (or one H2D upload if the input
// is still on the host), a host consumer the reference's memcpy -- so Splitter -> N x VFO
// moves the input over PCIe at most once instead of N times.
#pragma once
#include <cstring>
#include <vector>

#include "block.h"
#include "filter.h"

namespace dsp {

template <class T>
class Splitter : public generic_block<Splitter<T>> {
    using base = generic_block<Splitter<T>>;

public:
    Splitter() {}
    Splitter(stream<T>* in) { init(in); }
    ~Splitter() {
        const bool live = base::running;
        base::stop();
        if (live && _in) { _in->releaseConsumer(); }
    }

    void init(stream<T>* in) {
        _in = in;
        base::registerInput(_in);
        _in->claimConsumer(true, true);    // it can forward a device-resident block as is, with copies queued behind the producer's kernel (stream.h)
    }

    void setInput(stream<T>* in) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterInput(_in);
        _in->releaseConsumer();
        _in = in;
        _in->claimConsumer(true, true);
        base::registerInput(_in);
        base::tempStart();
    }

    void bindStream(stream<T>* s) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        out.push_back(s);
        base::registerOutput(s);
        base::tempStart();
    }

    void unbindStream(stream<T>* s) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        base::unregisterOutput(s);
        out.erase(std::remove(out.begin(), out.end(), s), out.end());
        base::tempStart();
    }

private:
    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        const int dev = detail::hipDeviceForBlocks();
        const size_t bytes = (size_t)count * sizeof(T);
        bool hostCopyValid = !_in->readOnDevice;
        const int inLink = _in->linkIn();
        for (stream<T>* s : out) {
            const bool toDev = s->consumerTakesDevice && s->ensureDevice(dev);
            int rc = 0, outLink = QDSP_HIP_LINK_HOST;
            if (toDev && _in->readOnDevice) {
                // device to device; towards a pipelined consumer the copy is only queued (behind the producer's kernel,
                // ahead of the consumer's), all of them before this block flushes its input
                outLink = s->linkOut(true);
                rc = qdsp_hip_memcpy_d2d_link(dev, s->devWriteBuf, _in->devReadBuf, bytes, inLink, outLink);
            } else if (toDev) {
                outLink = QDSP_HIP_LINK_DEVICE;
                rc = qdsp_hip_memcpy_h2d(dev, s->devWriteBuf, _in->readBuf, bytes);
            } else {
                if (!hostCopyValid) {  // a host consumer behind a device-resident input: one download, reused
                    rc = qdsp_hip_memcpy_d2h_link(dev, _in->readBuf, _in->devReadBuf, bytes, inLink);
                    hostCopyValid = rc == 0;
                }
                if (rc == 0) { memcpy(s->writeBuf, _in->readBuf, bytes); }
            }
            if (rc != 0) { _in->flush(); return detail::hipBlockFail("Splitter::run", rc); }
            s->markWritten(outLink);
            if (!s->swap(count)) { return -1; }
        }
        _in->flush();
        return count;
    }

    stream<T>* _in = nullptr;
    std::vector<stream<T>*> out;
};

}  // namespace dsp
