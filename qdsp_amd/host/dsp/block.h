// dsp/block.h -- block runtime: one worker thread per block looping on run().
//
// Public surface of the reference kept as is (src/dsp/block.h:13-133 generic_block,
// :135-208 generic_hier_block): start()/stop()/calcOutSize()/run(), and -- reachable by the
// derived BLOCK through `friend BLOCK` -- registerInput/registerOutput/unregister*,
// tempStart/tempStop and ctrlMtx.  Semantics:
//   start(): spawn the worker, which calls run() until it returns < 0        (block.h:55-57)
//   stop():  set the stop flag on every registered stream (inputs: reader side, outputs:
//            writer side) so a blocked read()/swap() returns, join, clear     (block.h:87-106)
//   tempStop()/tempStart(): the same, for live reconfiguration under ctrlMtx  (block.h:108-120)
#pragma once
#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "stream.h"
#include "types.h"

namespace dsp {

class generic_unnamed_block {
public:
    virtual ~generic_unnamed_block() {}
    virtual void start() {}
    virtual void stop() {}
    virtual int calcOutSize(int inSize) { return inSize; }
    virtual int run() { return -1; }
};

namespace detail {
// start/stop bookkeeping shared by leaf blocks and hierarchical blocks: `running` under ctrlMtx,
// and the temporary stop used while a live graph is re-plumbed.  What "on" and "off" mean
// (spawn/join a worker, or fan out to children) is supplied by the derived class.
class lifecycle {
protected:
    virtual ~lifecycle() {}
    virtual void doStart() = 0;
    virtual void doStop() = 0;

    void powerOn() {
        std::lock_guard<std::mutex> guard(ctrlMtx);
        if (running) { return; }
        running = true;
        doStart();
    }

    void powerOff() {
        std::lock_guard<std::mutex> guard(ctrlMtx);
        if (!running) { return; }
        doStop();
        running = false;
    }

    // callers hold ctrlMtx (reference: block.h:108-120)
    void tempStart() {
        if (!tempStopped) { return; }
        doStart();
        tempStopped = false;
    }

    void tempStop() {
        if (!running || tempStopped) { return; }
        doStop();
        tempStopped = true;
    }

    bool running = false;
    bool tempStopped = false;
    std::mutex ctrlMtx;
};
}  // namespace detail

template <class BLOCK>
class generic_block : public generic_unnamed_block, protected detail::lifecycle {
public:
    virtual void init() {}

    virtual ~generic_block() { stop(); }

    void start() override { powerOn(); }
    void stop() override { powerOff(); }

    int calcOutSize(int inSize) override { return inSize; }

    int run() override = 0;

    friend BLOCK;

private:
    static void drop(std::vector<untyped_steam*>& v, untyped_steam* s) { v.erase(std::remove(v.begin(), v.end(), s), v.end()); }

    void registerInput(untyped_steam* s) { inputs.push_back(s); }
    void unregisterInput(untyped_steam* s) { drop(inputs, s); }
    void registerOutput(untyped_steam* s) { outputs.push_back(s); }
    void unregisterOutput(untyped_steam* s) { drop(outputs, s); }

    // the worker: run() until it asks to end (reference: block.h:55-57)
    void doStart() override {
        worker = std::thread([this] { while (this->run() >= 0) {} });
    }

    // wake whatever the worker is blocked in, join it, re-arm the streams (block.h:87-106)
    void doStop() override {
        for (untyped_steam* s : inputs) { s->stopReader(); }
        for (untyped_steam* s : outputs) { s->stopWriter(); }
        if (worker.joinable()) { worker.join(); }
        for (untyped_steam* s : inputs) { s->clearReadStop(); }
        for (untyped_steam* s : outputs) { s->clearWriteStop(); }
    }

    std::vector<untyped_steam*> inputs;
    std::vector<untyped_steam*> outputs;
    std::thread worker;
};

// A block made of blocks: start/stop fan out to the registered children.
template <class BLOCK>
class generic_hier_block : protected detail::lifecycle {
public:
    virtual void init() {}

    virtual ~generic_hier_block() { stop(); }

    virtual void start() { powerOn(); }
    virtual void stop() { powerOff(); }

    virtual int calcOutSize(int inSize) { return inSize; }

    friend BLOCK;

private:
    void registerBlock(generic_unnamed_block* b) { blocks.push_back(b); }
    void unregisterBlock(generic_unnamed_block* b) { blocks.erase(std::remove(blocks.begin(), blocks.end(), b), blocks.end()); }

    void doStart() override { for (auto* b : blocks) { b->start(); } }
    void doStop() override { for (auto* b : blocks) { b->stop(); } }

    std::vector<generic_unnamed_block*> blocks;
};

}  // namespace dsp
