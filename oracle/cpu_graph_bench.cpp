// cpu_graph_bench -- the CPU path as the reference runs it: ONE block-graph worker thread per block,
// source -> FIR (or PolyphaseResampler) -> NullSink over dsp::stream<T> hand-offs
// (reference: src/dsp/block.h:55-57,83-85 one std::thread per block looping on run(); src/dsp/filter.h:51-74,
// src/dsp/resampling.h:99-132 for what run() does; src/dsp/sink.h:96-132 NullSink).
//
// TEST INFRASTRUCTURE ONLY (lives under oracle/): the filter block below is a CPU block whose arithmetic is the
// oracle's restatement of the reference loop (oracle_fir_cf32 / oracle_resamp_cf32, qdsp_oracle.c), run on the
// block runtime mirror of qdsp_amd/host/dsp (stream.h / block.h / source.h / sink.h: the reference's stream and
// block API).  bench.py's cpu_baseline leg runs it and reports `value_1thread_graph`; nothing in the product uses it.
//
//   cpu_graph_bench <fir|decim> <taps.f32> <decim> <acc: 0 generic order, 3 SIMD lanes> <block> <seconds>
// prints one line:  cpu_graph <workload> acc=<a> block=<b> samples=<n> seconds=<s> msps=<r>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include <dsp/block.h>
#include <dsp/sink.h>
#include <dsp/source.h>

extern "C" {
long oracle_fir_cf32(const float* taps, int ntaps, float* hist, const float* in, long count, float* out, int acc);
long oracle_resamp_cf32(const float* taps, int ntaps, int interp, int decim, float* hist, const float* in, long count, float* out, int acc);
void oracle_synth_iq(float* out, long first_sample, long count, unsigned int seed);
}

using namespace dsp;

// FIR<complex_t> / PolyphaseResampler<complex_t> with interp 1, on the CPU: read -> filter -> flush -> swap
class CpuFilter : public generic_block<CpuFilter> {
    using base = generic_block<CpuFilter>;

public:
    CpuFilter(stream<complex_t>* in, std::vector<float> t, int decim, int acc) : _in(in), taps(std::move(t)), _decim(decim), _acc(acc) {
        hist.assign(2 * taps.size(), 0.0f);   // FIR: ntaps samples (filter.h:28); resampler interp 1: tapsPerPhase = ntaps
        base::registerInput(_in);
        base::registerOutput(&out);
    }
    ~CpuFilter() { base::stop(); }

    int run() override {
        const int count = _in->read();
        if (count < 0) { return -1; }
        long n;
        if (_decim <= 1) {
            n = oracle_fir_cf32(taps.data(), (int)taps.size(), hist.data(), reinterpret_cast<const float*>(_in->readBuf), count,
                                reinterpret_cast<float*>(out.writeBuf), _acc);
        } else {
            n = oracle_resamp_cf32(taps.data(), (int)taps.size(), 1, _decim, hist.data(), reinterpret_cast<const float*>(_in->readBuf), count,
                                   reinterpret_cast<float*>(out.writeBuf), _acc);
        }
        _in->flush();
        if (n < 0) { return -1; }
        done.fetch_add(count, std::memory_order_relaxed);
        if (!out.swap((int)n)) { return -1; }
        return count;
    }

    stream<complex_t> out;
    std::atomic<long> done{0};

private:
    stream<complex_t>* _in;
    std::vector<float> taps, hist;
    int _decim, _acc;
};

struct Feed {
    std::vector<complex_t> block;
    static int pull(complex_t* dst, void* ctx) {
        Feed* f = static_cast<Feed*>(ctx);
        memcpy(dst, f->block.data(), f->block.size() * sizeof(complex_t));   // what a file / SDR source does per block
        return (int)f->block.size();
    }
};

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: cpu_graph_bench <fir|decim> <taps.f32> <decim> <acc> <block> <seconds>\n"); return 2; }
    const std::string kind = argv[1];
    std::ifstream tf(argv[2], std::ios::binary | std::ios::ate);
    if (!tf) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    std::vector<float> taps((size_t)tf.tellg() / sizeof(float));
    tf.seekg(0);
    tf.read(reinterpret_cast<char*>(taps.data()), (std::streamsize)(taps.size() * sizeof(float)));
    const int decim = atoi(argv[3]), acc = atoi(argv[4]), block = atoi(argv[5]);
    const double seconds = atof(argv[6]);
    if (taps.empty() || block <= 0 || block > STREAM_BUFFER_SIZE || decim < 1) { fprintf(stderr, "bad arguments\n"); return 2; }

    Feed feed;
    feed.block.resize((size_t)block);
    oracle_synth_iq(reinterpret_cast<float*>(feed.block.data()), 0, block, 1234u);
    HandlerSource<complex_t> src(Feed::pull, &feed);
    CpuFilter filt(&src.out, taps, kind == "fir" ? 1 : decim, acc);
    NullSink<complex_t> sink(&filt.out);
    sink.start();
    filt.start();
    src.start();
    // let the graph fill, then count what the filter block gets through in `seconds`
    while (filt.done.load() < 2L * block) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
    const long n0 = filt.done.load();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const long n1 = filt.done.load();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    src.stop();
    filt.stop();
    sink.stop();
    printf("cpu_graph %s acc=%d block=%d samples=%ld seconds=%.3f msps=%.3f\n", kind.c_str(), acc, block, n1 - n0, sec, (double)(n1 - n0) / sec / 1e6);
    return 0;
}
