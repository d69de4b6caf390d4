// dsp/vfo_bank.h -- the hand-shake between a Splitter and the VFO cores behind its output links that lets N identical
// VFOs (same taps / interp / decim, any offsets) run as ONE batched launch per block
// (qdsp_hip_chan_cf32_process_links: resamp_any_batch_kernel, blockIdx.y = channel) instead of N copies + N kernels.
//
// The reference's channelizer shape is exactly this graph: Splitter -> N x VFO (src/dsp/routing.h:47-57 memcpy fan-out,
// src/dsp/vfo.h:19-36).  Nothing about the graph changes for its author: the blocks, their threads and their stream
// protocol stay; while a bank is active the links from the Splitter carry token blocks (a count, no copy) and each VFO
// worker only passes on the output block the bank has already computed into its `out` stream.
//
// Lifetime: descriptors and the control block are shared_ptr-owned by both sides, so neither end ever looks at freed
// memory if the other is destroyed first; a destroyed / reconfigured member raises `broken` and the Splitter takes the
// bank down at its next block (the VFOs then run their own kernels again, from their own -- stale -- filter state).
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "stream.h"
#include "types.h"

namespace dsp {
namespace detail {

struct vfo_bank_ctl {
    std::mutex m;                       // guards `bank` against retunes from control threads
    void* bank = nullptr;               // qdsp_hip_chan_cf32 handle, owned by the Splitter
    std::atomic<bool> broken{false};    // a member was reconfigured or destroyed: rebuild or give up
};

struct vfo_bank_member {
    // the design the core runs: written by the core (init / configure / retune, any thread), read by the Splitter's
    // worker when it builds a bank -- both under designMtx
    std::mutex designMtx;
    std::vector<float> taps;
    int interp = 1, decim = 1;
    float dRe = 1.0f, dIm = 0.0f;
    stream<complex_t>* out = nullptr;   // the core's output stream
    void* handle = nullptr;             // the core's own xlate_fir_decim_cf32 handle: idle while banked, takes the channel's state back afterwards
    std::atomic<bool> alive{true};
    // while banked
    std::shared_ptr<vfo_bank_ctl> ctl;  // nullptr = not banked: the core runs its own kernel
    int index = 0;
    // per block: written by the Splitter before it swaps the token block in, read by the core after read()
    int outCount = 0;
    int outLink = QDSP_HIP_LINK_HOST;
    void* evt = nullptr;
};

}  // namespace detail
}  // namespace dsp
