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
#include "vfo_bank.h"

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
        dropBank();
        for (void*& e : bankEvt) { if (e) { qdsp_hip_event_destroy(e); e = nullptr; } }
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
        dropBank();
        bankOff = false;
        out.push_back(s);
        base::registerOutput(s);
        base::tempStart();
    }

    void unbindStream(stream<T>* s) {
        std::lock_guard<std::mutex> lck(base::ctrlMtx);
        base::tempStop();
        dropBank();
        bankOff = false;
        base::unregisterOutput(s);
        out.erase(std::remove(out.begin(), out.end(), s), out.end());
        base::tempStart();
    }

private:
    // ---- N identical VFOs behind the outputs: one batched launch per block (vfo_bank.h) ------------------------------
    void dropBank() {
        // every running VFO gets its channel's NCO phase and filter history back: it carries on without a glitch
        if (ctl && ctl->bank) {
            for (auto& m : members) {
                std::lock_guard<std::mutex> lk(m->designMtx);       // (a VFO being destroyed clears `handle` under this mutex)
                if (m->alive.load() && m->handle) { qdsp_hip_chan_cf32_move_channel_state(ctl->bank, m->index, m->handle, 1); }
            }
        }
        for (auto& m : members) { std::atomic_store(&m->ctl, std::shared_ptr<detail::vfo_bank_ctl>()); }
        members.clear();
        if (ctl) {
            std::lock_guard<std::mutex> lk(ctl->m);
            if (ctl->bank) { qdsp_hip_chan_cf32_destroy(ctl->bank); }
            ctl->bank = nullptr;
        }
        ctl.reset();
        // (the two completion events stay until the Splitter goes: a consumer downstream of a VFO may still be waiting
        // on the one that travelled with the last banked block)
    }

    // all consumers idle (every link flushed): see whether they are N >= 2 live VFO cores of one design
    bool buildBank() {
        if (out.size() < 2) { return false; }
        const char* no = getenv("QDSP_HIP_NO_VFO_BANK");
        if (no && atoi(no)) { return false; }
        std::vector<std::shared_ptr<detail::vfo_bank_member>> ms;
        for (stream<T>* s : out) {
            auto m = s->bankMember;
            if (!m || !m->alive.load() || std::atomic_load(&m->ctl)) { return false; }
            ms.push_back(m);
        }
        // a consistent snapshot of every member's design
        std::vector<float> taps0, re, im;
        int interp0 = 0, decim0 = 0;
        for (size_t i = 0; i < ms.size(); i++) {
            std::lock_guard<std::mutex> lk(ms[i]->designMtx);
            if (!ms[i]->out) { return false; }
            if (i == 0) { taps0 = ms[i]->taps; interp0 = ms[i]->interp; decim0 = ms[i]->decim; }
            else if (ms[i]->taps != taps0 || ms[i]->interp != interp0 || ms[i]->decim != decim0) { return false; }
            re.push_back(ms[i]->dRe);
            im.push_back(ms[i]->dIm);
        }
        if (taps0.empty()) { return false; }
        const int dev = detail::hipDeviceForBlocks();
        void* h = nullptr;
        if (qdsp_hip_chan_cf32_create(&h, dev, taps0.data(), (int)taps0.size(), interp0, decim0, (int)ms.size(), re.data(), im.data(), 0) != 0) {
            return false;
        }
        for (void*& e : bankEvt) {
            if (!e && qdsp_hip_event_create(dev, &e) != 0) {
                e = nullptr;
                qdsp_hip_chan_cf32_destroy(h);
                return false;
            }
        }
        // a bank built mid-stream continues where the VFOs' own kernels stopped
        for (size_t i = 0; i < ms.size(); i++) {
            std::lock_guard<std::mutex> lk(ms[i]->designMtx);
            if (ms[i]->alive.load() && ms[i]->handle) { qdsp_hip_chan_cf32_move_channel_state(h, (int)i, ms[i]->handle, 0); }
        }
        ctl = std::make_shared<detail::vfo_bank_ctl>();
        ctl->bank = h;
        members = ms;
        for (size_t i = 0; i < members.size(); i++) {
            members[i]->index = (int)i;
            std::atomic_store(&members[i]->ctl, ctl);
        }
        // a retune that raced the build: push every member's current increment again (they are applied between blocks)
        for (auto& m : members) {
            std::lock_guard<std::mutex> lk(m->designMtx);
            qdsp_hip_chan_cf32_set_phase_inc(h, m->index, m->dRe, m->dIm);
        }
        return true;
    }

    // One block through the bank.  true: done (every link got its token block); false: not banked -- copy the block out
    // as the reference does.  -1 in `stop`: a link is being stopped.
    bool runBank(int count, bool& stop) {
        stop = false;
        if (bankOff) {
            // not sticky for good: a member that was mid-configure, or whose design only matched the others' a
            // little later, gets another look every kBankRetryBlocks blocks
            if (++bankOffBlocks < kBankRetryBlocks) { return false; }
            bankOff = false;
            bankOffBlocks = 0;
        }
        if (ctl && ctl->broken.load()) {
            for (stream<T>* s : out) { if (!s->waitFlushed()) { stop = true; return false; } }
            dropBank();
        }
        if (!ctl) {
            for (stream<T>* s : out) { if (!s->waitFlushed()) { stop = true; return false; } }
            if (!buildBank()) { bankOff = true; return false; }
        }
        const int dev = detail::hipDeviceForBlocks();
        const size_t n = members.size();
        std::vector<void*> outs(n);
        std::vector<int> links(n);
        void* evt = bankEvt[blockNo & 1];
        for (size_t i = 0; i < n; i++) {
            // member i has finished the previous block -- flushed its token AFTER swapping its output -- so the write
            // buffer of its out stream is free
            if (!out[i]->waitFlushed()) { stop = true; return false; }
            stream<complex_t>* o = nullptr;
            {
                std::lock_guard<std::mutex> lk(members[i]->designMtx);
                o = members[i]->alive.load() ? members[i]->out : nullptr;
            }
            if (!o) {               // destroyed between the `broken` test above and here: every link is idle up to i, none has a token yet
                dropBank();
                return false;
            }
            const bool outDev = o->consumerTakesDevice && o->ensureDevice(dev);
            outs[i] = outDev ? static_cast<void*>(o->devWriteBuf) : static_cast<void*>(o->writeBuf);
            links[i] = outDev ? o->linkOut(true) : QDSP_HIP_LINK_HOST_DEFERRED;
        }
        const void* src = _in->readOnDevice ? static_cast<const void*>(_in->devReadBuf) : static_cast<const void*>(_in->readBuf);
        const long long rc = qdsp_hip_chan_cf32_process_links(ctl->bank, src, _in->linkIn(), count, outs.data(), links.data(), evt);
        if (rc < 0) {
            // not servable this way (pageable / oversized host buffer, taps beyond LDS): back to one kernel per VFO for good
            dropBank();
            bankOff = true;
            return false;
        }
        for (size_t i = 0; i < n; i++) {
            members[i]->outCount = (int)rc;
            members[i]->outLink = links[i];
            members[i]->evt = links[i] == QDSP_HIP_LINK_HOST_DEFERRED ? evt : nullptr;
        }
        blockNo++;
        for (stream<T>* s : out) {
            s->markWritten(QDSP_HIP_LINK_HOST);
            s->markToken();                                         // token block: the count, no samples
            if (!s->swap(count)) { stop = true; return true; }
        }
        return true;
    }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        if constexpr (std::is_same<T, complex_t>::value) {
            bool stop = false;
            const bool done = runBank(count, stop);
            if (done || stop) {
                _in->flush();
                return stop ? -1 : count;
            }
        }
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
    // bank state (complex_t only)
    std::shared_ptr<detail::vfo_bank_ctl> ctl;
    std::vector<std::shared_ptr<detail::vfo_bank_member>> members;
    void* bankEvt[2] = {nullptr, nullptr};
    unsigned blockNo = 0;
    bool bankOff = false;     // decided for now: these consumers cannot be banked
    unsigned bankOffBlocks = 0;
    static constexpr unsigned kBankRetryBlocks = 64;
};

}  // namespace dsp
