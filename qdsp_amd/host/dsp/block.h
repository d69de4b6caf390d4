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

template <class BLOCK>
class generic_block : public generic_unnamed_block {
public:
    virtual void init() {}

    virtual ~generic_block() { stop(); }

    void start() override {
        std::lock_guard<std::mutex> lk(ctrlMtx);
        if (running) { return; }
        running = true;
        doStart();
    }

    void stop() override {
        std::lock_guard<std::mutex> lk(ctrlMtx);
        if (!running) { return; }
        doStop();
        running = false;
    }

    int calcOutSize(int inSize) override { return inSize; }

    int run() override = 0;

    friend BLOCK;

private:
    void registerInput(untyped_steam* s) { inputs.push_back(s); }
    void unregisterInput(untyped_steam* s) { inputs.erase(std::remove(inputs.begin(), inputs.end(), s), inputs.end()); }
    void registerOutput(untyped_steam* s) { outputs.push_back(s); }
    void unregisterOutput(untyped_steam* s) { outputs.erase(std::remove(outputs.begin(), outputs.end(), s), outputs.end()); }

    virtual void doStart() {
        worker = std::thread([this] { while (this->run() >= 0) {} });
    }

    virtual void doStop() {
        for (untyped_steam* s : inputs) { s->stopReader(); }
        for (untyped_steam* s : outputs) { s->stopWriter(); }
        if (worker.joinable()) { worker.join(); }
        for (untyped_steam* s : inputs) { s->clearReadStop(); }
        for (untyped_steam* s : outputs) { s->clearWriteStop(); }
    }

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

    std::vector<untyped_steam*> inputs;
    std::vector<untyped_steam*> outputs;
    bool running = false;
    bool tempStopped = false;
    std::thread worker;

protected:
    std::mutex ctrlMtx;
};

// A block made of blocks: start/stop fan out to the registered children.
template <class BLOCK>
class generic_hier_block {
public:
    virtual void init() {}

    virtual ~generic_hier_block() { stop(); }

    virtual void start() {
        std::lock_guard<std::mutex> lk(ctrlMtx);
        if (running) { return; }
        running = true;
        doStart();
    }

    virtual void stop() {
        std::lock_guard<std::mutex> lk(ctrlMtx);
        if (!running) { return; }
        doStop();
        running = false;
    }

    virtual int calcOutSize(int inSize) { return inSize; }

    friend BLOCK;

private:
    void registerBlock(generic_unnamed_block* b) { blocks.push_back(b); }
    void unregisterBlock(generic_unnamed_block* b) { blocks.erase(std::remove(blocks.begin(), blocks.end(), b), blocks.end()); }

    virtual void doStart() { for (auto* b : blocks) { b->start(); } }
    virtual void doStop() { for (auto* b : blocks) { b->stop(); } }

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

    std::vector<generic_unnamed_block*> blocks;
    bool tempStopped = false;
    bool running = false;

protected:
    std::mutex ctrlMtx;
};

}  // namespace dsp
