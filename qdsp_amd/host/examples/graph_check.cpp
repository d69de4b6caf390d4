// graph_check -- drives the C++ block graph (qdsp_amd/host/dsp) from the command line so
// the tests can compare what comes out of a real source -> block -> sink graph with the
// CPU oracle.  Not part of the product; a harness.
//
//   graph_check taps   <out.f32>                      window designers -> tap tables (no GPU)
//   graph_check stream                                stream/block protocol self-test (no GPU)
//   graph_check fail                                  a block whose GPU handle could not be created: its worker ends, the failure is
//                                                     visible through dsp::hipBlockErrors() / hipBlockLastError()
//   graph_check fir    <in.cf32> <out.cf32> <block> <taps.f32>
//   graph_check fir63  <in.cf32> <out.cf32> <block>   BlackmanWindow(0.1 fs, 4 fs/63, fs=1)
//   graph_check firf   <in.f32>  <out.f32>  <block> <taps.f32>          (FIR<float>)
//   graph_check firrrc <in.cf32> <out.cf32> <block> <tapCount> <sampleRate> <baudRate> <alpha>
//                      FIR<complex_t>(RRCTaps): the matched filter of the reference's PSKDemod (demodulator.h:586-587); an EVEN
//                      tapCount makes RRCTaps design tapCount + 1 taps (window.h:183) of which the FIR keeps the first tapCount
//   graph_check firbp  <in.f32>  <out.f32>  <block> <cutoff> <transWidth> <offset> <sampleRate>
//                      FIR<float>(BlackmanBandpassWindow): the pilot filter of the reference's StereoFMDemod (demodulator.h:216-217)
//   graph_check resamp <in.cf32> <out.cf32> <block> <inSR> <outSR> <cutoff> <trans>
//   graph_check resampst <in.cf32> <out.cf32> <block> <inSR> <outSR> <cutoff> <trans>
//                      the same through PolyphaseResampler<stereo_t> (resampling.h:120: a float pair with real taps)
//   graph_check xlate  <in.cf32> <out.cf32> <block> <sampleRate> <freq>
//   graph_check vfo    <in.cf32> <out.cf32> <block> <offset> <inSR> <outSR> <bw>
//   graph_check wavfir <in.wav>  <out.cf32> <block>   config 1: int16 IQ WAV -> 63-tap FIR
//   graph_check sine   <out.cf32> <blockSize> <nblocks> <sampleRate> <freq> [fir_taps.f32]
//                      SineSource -> (optional FIR) -> sink
//   graph_check chain  <in.cf32> <out.cf32> <block> <taps.f32> <sampleRate> <freq> <inSR> <outSR>
//                      source -> FrequencyXlator -> FIR -> PolyphaseResampler -> sink: three GPU
//                      blocks in a row, the two links between them device-resident
//   graph_check math   <in.cf32> <out.cf32> <block> add|sub|mul <sampleRate> <freq>
//                      source -> Splitter -> { FrequencyXlator, identity } -> Add | Substract | Multiply -> sink
//   graph_check bench  vfo|chain|split<n>|hostfir|hostvfo <blockSize> <nblocks> <inSR> <outSR>
//                      throughput of a live graph: SineSource -> VFO -> sink, or SineSource -> FrequencyXlator ->
//                      PolyphaseResampler -> sink (the same two blocks unfused, device-resident link between them);
//                      prints input Msamples/s and microseconds per block.  hostfir / hostvfo: a HOST source (HandlerSource whose handler
//                      leaves the pinned block as it is) -> FIR<complex_t> (256 taps) / VFO -> host sink: what an unmodified qdsp
//                      graph with host-resident streams gets -- the PCIe link both ways
//   graph_check split  <in.cf32> <out_prefix> <block> <n> <inSR> <outSR> <bw>
//                      source -> Splitter -> n x VFO(offset_i = (i - (n-1)/2) * inSR/n) -> sinks
//   graph_check splitretune <in.cf32> <out_prefix> <block> <n> <inSR> <outSR> <bw> <K> <newOffset> [reconf]
//                      as split, but before block K is fed (and after every sink has blocks 0..K-1) VFO 1 is retuned to
//                      newOffset (setOffset, vfo.h:78-82: the bank's channel follows); with `reconf` VFO 0 is given a new
//                      bandwidth instead (setBandwidth: the bank is taken down, every VFO runs its own kernel again)
//   graph_check mulsplit <in.cf32> <out_prefix> <block> <n> <inSR> <outSR> <bw>
//                      source -> Splitter (host-fed) -> Multiply(x, x) -> Splitter -> n x VFO -> sinks: the second
//                      Splitter's input comes from a producer OUTSIDE the library's pipelined stream, so its
//                      device-to-device copies must have read the block before it is flushed
//   graph_check shard  <in.cf32> <out.cf32> <chunk> <taps.f32> [<rank> <world> <idfile>]
//                      the time-sharded path from C++ (include/qdsp_hip.h "ring"): this rank filters chunks rank, rank + world,
//                      ... of the stream with FIR<complex_t>'s engine, every chunk's history = the tail of the chunk before it,
//                      delivered by qdsp_hip_ring_post / _complete (RCCL send / recv, posted one step ahead); rank 0 writes the
//                      RCCL id to <idfile>, the others wait for it.  Default: one rank, its own ring neighbour.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <dsp/filter.h>
#include <dsp/math.h>
#include <dsp/processing.h>
#include <dsp/resampling.h>
#include <dsp/routing.h>
#include <dsp/sink.h>
#include <dsp/source.h>
#include <dsp/vfo.h>
#include <wavreader.h>

using namespace dsp;

template <class T> static std::vector<T> readAll(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    std::vector<T> v(bytes / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
    return v;
}

// taps handed in through the reference's own extension point (window.h:7-11)
class FileTaps : public filter_window::generic_window {
public:
    explicit FileTaps(const char* path) : t(readAll<float>(path)) {}
    int getTapCount() override { return (int)t.size(); }
    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        for (int i = 0; i < tapCount; i++) { taps[i] = t[i] * factor; }
    }
private:
    std::vector<float> t;
};

template <class T> struct Feed {
    std::vector<T> data;
    size_t pos = 0;
    int block = 1;
    static int pull(T* dst, void* ctx) {
        Feed* f = static_cast<Feed*>(ctx);
        if (f->pos >= f->data.size()) { return -1; }
        const size_t n = std::min<size_t>((size_t)f->block, f->data.size() - f->pos);
        memcpy(dst, f->data.data() + f->pos, n * sizeof(T));
        f->pos += n;
        return (int)n;
    }
};

template <class T> struct Collect {
    std::vector<T> data;
    std::atomic<long> blocks{0};
    static void push(T* src, int count, void* ctx) {
        Collect* c = static_cast<Collect*>(ctx);
        c->data.insert(c->data.end(), src, src + count);
        c->blocks++;
    }
};

// source -> BLOCK -> sink; waits until every input block has come out the far end.
template <class T, class MAKE> static int runGraph(const char* inPath, const char* outPath, int block, MAKE make) {
    Feed<T> feed;
    feed.data = readAll<T>(inPath);
    feed.block = block;
    const long nblocks = (long)((feed.data.size() + block - 1) / block);
    Collect<T> sinkData;
    HandlerSource<T> src(Feed<T>::pull, &feed);
    auto* blk = make(&src.out);
    HandlerSink<T> sink(&blk->out, Collect<T>::push, &sinkData);
    sink.start();
    blk->start();
    src.start();
    const auto t0 = std::chrono::steady_clock::now();
    while (sinkData.blocks.load() < nblocks) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "graph timed out\n"); return 3; }
    }
    src.stop();
    blk->stop();
    sink.stop();
    delete blk;
    std::ofstream o(outPath, std::ios::binary);
    o.write(reinterpret_cast<const char*>(sinkData.data.data()), (std::streamsize)(sinkData.data.size() * sizeof(T)));
    printf("graph ok: %zu in, %zu out, %ld blocks\n", feed.data.size(), sinkData.data.size(), nblocks);
    return 0;
}

static int dumpTaps(const char* outPath) {
    std::ofstream o(outPath, std::ios::binary);
    auto dump = [&](filter_window::generic_window& w, float factor) {
        const int n = w.getTapCount();
        std::vector<float> t((size_t)n + 1, 0.0f);
        w.createTaps(t.data(), n, factor);
        const float fn = (float)n;
        o.write(reinterpret_cast<const char*>(&fn), 4);
        o.write(reinterpret_cast<const char*>(t.data()), 4 * n);
    };
    filter_window::BlackmanWindow a(0.1f, 4.0f / 63.0f, 1.0f);
    dump(a, 1.0f);
    filter_window::BlackmanBandpassWindow b(0.05f, 4.0f / 63.0f, 0.2f, 1.0f);
    dump(b, 1.0f);
    RRCTaps c(31, 4.0f, 1.0f, 0.35f);
    dump(c, 1.0f);
    // the VFO's design for 2.4 Msps -> 240 ksps, bw 200 kHz (vfo.h:26-33): 97 taps at gain 1
    filter_window::BlackmanWindow d(100e3f, 100e3f, 2.4e6f);
    dump(d, 1.0f);
    filter_window::BlackmanWindow e(0.1f, 4.0f / 63.0f, 1.0f);
    dump(e, 3.0f);  // interp gain
    return 0;
}

// stream / block protocol, no GPU involved
static int streamSelfTest() {
    // 1. data integrity + exactly-one-in-flight over many blocks
    {
        stream<int> s;
        std::atomic<bool> bad{false};
        std::thread prod([&] {
            for (int b = 0; b < 2000; b++) {
                const int n = 1 + (b * 37) % 5000;
                for (int i = 0; i < n; i++) { s.writeBuf[i] = b * 10000 + i; }
                if (!s.swap(n)) { bad = true; return; }
            }
        });
        for (int b = 0; b < 2000; b++) {
            const int n = s.read();
            if (n != 1 + (b * 37) % 5000) { bad = true; }
            for (int i = 0; i < n; i++) { if (s.readBuf[i] != b * 10000 + i) { bad = true; } }
            s.flush();
        }
        prod.join();
        if (bad) { printf("FAIL stream integrity\n"); return 1; }
    }
    // 2. stopReader wakes a blocked read with -1; stopWriter wakes a blocked swap with false
    {
        stream<float> s;
        int r = 0;
        std::thread t([&] { r = s.read(); });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        s.stopReader();
        t.join();
        if (r != -1) { printf("FAIL stopReader\n"); return 1; }
        s.clearReadStop();
        if (!s.swap(3)) { printf("FAIL first swap\n"); return 1; }  // slot free -> immediate
        bool ok = true;
        std::thread w([&] { ok = s.swap(4); });                     // blocks: consumer never flushed
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        s.stopWriter();
        w.join();
        if (ok) { printf("FAIL stopWriter\n"); return 1; }
        s.clearWriteStop();
        if (s.read() != 3) { printf("FAIL pending size\n"); return 1; }
    }
    // 3. generic_block lifecycle: source -> sink, stop() joins both, restart works
    {
        struct Ctx { int n = 0; } ctx;
        HandlerSource<float> src([](float* d, void* c) { static_cast<Ctx*>(c)->n++; d[0] = 1.0f; return 1; }, &ctx);
        std::atomic<long> got{0};
        HandlerSink<float> sink(&src.out, [](float*, int n, void* c) { *static_cast<std::atomic<long>*>(c) += n; }, &got);
        for (int round = 0; round < 2; round++) {
            sink.start();
            src.start();
            while (got.load() < 1000 * (round + 1)) { std::this_thread::yield(); }
            src.stop();
            sink.stop();
        }
        NullSink<float> ns(&src.out);
        ns.start();
        src.start();
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        src.stop();
        ns.stop();
    }
    // 4. window parity facts the kernels rely on
    {
        filter_window::BlackmanWindow w(0.1f, 4.0f / 63.0f, 1.0f);
        if (w.getTapCount() != 63) { printf("FAIL tap count %d\n", w.getTapCount()); return 1; }
    }
#ifndef QDSP_GRAPH_CHECK_BIG_BLOCKS
    static_assert(STREAM_BUFFER_SIZE == 1000000, "stream capacity is part of the contract (src/dsp/stream.h:7)");
#endif
    printf("stream/block self-test ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: see header of graph_check.cpp\n"); return 2; }
    const std::string mode = argv[1];
    if (mode == "taps" && argc >= 3) { return dumpTaps(argv[2]); }
    if (mode == "stream") { return streamSelfTest(); }
    if (mode == "fail") {
        // an empty tap table cannot make a handle (QDSP_HIP_EINVAL): init reports it, the worker's first run() ends the block
        struct NoTaps : filter_window::generic_window {
            int getTapCount() override { return 0; }
            void createTaps(float*, int, float) override {}
        } none;
        const long before = hipBlockErrors();
        HandlerSource<complex_t> src([](complex_t* d, void*) { d[0] = complex_t{1.0f, 0.0f}; return 1; }, nullptr);
        FIR<complex_t> fir(&src.out, &none);
        NullSink<complex_t> sink(&fir.out);
        sink.start();
        fir.start();
        src.start();
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        src.stop();
        fir.stop();
        sink.stop();
        const char* who = nullptr;
        const int code = hipBlockLastError(&who);
        printf("errors %ld code %d who %s\n", hipBlockErrors() - before, code, who ? who : "-");
        return (hipBlockErrors() > before && code != 0 && who) ? 0 : 1;
    }
    if (mode == "sine" && argc >= 7) {
        const int bs = atoi(argv[3]), nb = atoi(argv[4]);
        SineSource src(bs, (float)atof(argv[5]), (float)atof(argv[6]));
        Collect<complex_t> col;
        FileTaps* taps = argc >= 8 ? new FileTaps(argv[7]) : nullptr;
        FIR<complex_t>* fir = taps ? new FIR<complex_t>(&src.out, taps) : nullptr;
        HandlerSink<complex_t> sink(fir ? &fir->out : &src.out, Collect<complex_t>::push, &col);
        sink.start();
        if (fir) { fir->start(); }
        src.start();
        const auto t0 = std::chrono::steady_clock::now();
        while (col.blocks.load() < nb) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { fprintf(stderr, "sine graph timed out\n"); return 3; }
        }
        src.stop();
        if (fir) { fir->stop(); }
        sink.stop();
        col.data.resize((size_t)bs * nb);
        std::ofstream o(argv[2], std::ios::binary);
        o.write(reinterpret_cast<const char*>(col.data.data()), (std::streamsize)(col.data.size() * sizeof(complex_t)));
        delete fir;
        delete taps;
        printf("sine ok: %d blocks of %d\n", nb, bs);
        return 0;
    }
    if (mode == "bench" && argc >= 7) {
        const std::string kind = argv[2];
        const int bs = atoi(argv[3]), nb = atoi(argv[4]);
        const float inSR = (float)atof(argv[5]), outSR = (float)atof(argv[6]);
        if (bs <= 0 || bs > STREAM_BUFFER_SIZE || nb <= 0) { fprintf(stderr, "bad block size / count\n"); return 2; }
        struct Count {
            std::atomic<long> blocks{0}, samples{0};
            static void push(complex_t*, int count, void* ctx) {
                Count* c = static_cast<Count*>(ctx);
                c->samples += count;
                c->blocks++;
            }
        } cnt;
        SineSource src(bs, inSR, inSR * 0.01f);
        VFO* vfo = nullptr;
        FrequencyXlator<complex_t>* xl = nullptr;
        PolyphaseResampler<complex_t>* rs = nullptr;
        filter_window::BlackmanWindow win(outSR / 2.0f, outSR / 2.0f, inSR);
        stream<complex_t>* tail = nullptr;
        if (kind == "vfo") {
            vfo = new VFO(&src.out, inSR * 0.1f, inSR, outSR, outSR);
            tail = vfo->out;
        } else if (kind == "chain") {
            xl = new FrequencyXlator<complex_t>(&src.out, inSR, -inSR * 0.1f);
            rs = new PolyphaseResampler<complex_t>(&xl->out, &win, inSR, outSR);
            tail = &rs->out;
        } else if (kind.rfind("split", 0) == 0) {
            // SineSource -> Splitter -> n x VFO -> n sinks (the reference's channelizer shape); blocks counted on VFO 0
            const int n = atoi(kind.c_str() + 5) > 0 ? atoi(kind.c_str() + 5) : 4;
            Splitter<complex_t> split(&src.out);
            std::vector<stream<complex_t>*> legs;
            std::vector<VFO*> vfos;
            std::vector<HandlerSink<complex_t>*> sinks;
            Count others;
            for (int i = 0; i < n; i++) {
                legs.push_back(new stream<complex_t>());
                split.bindStream(legs.back());
                vfos.push_back(new VFO(legs.back(), ((float)i - (float)(n - 1) / 2.0f) * inSR / (float)n * 0.5f, inSR, outSR, outSR));
                sinks.push_back(new HandlerSink<complex_t>(vfos.back()->out, Count::push, i == 0 ? &cnt : &others));
            }
            for (auto* k : sinks) { k->start(); }
            for (auto* v : vfos) { v->start(); }
            split.start();
            src.start();
            auto wait_blocks = [&](long want) {
                const auto t0 = std::chrono::steady_clock::now();
                while (cnt.blocks.load() < want) {
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { return false; }
                }
                return true;
            };
            if (!wait_blocks(nb / 10 + 2)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
            const long b0 = cnt.blocks.load();
            const auto t0 = std::chrono::steady_clock::now();
            if (!wait_blocks(b0 + nb)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
            const long b1 = cnt.blocks.load();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            src.stop();
            split.stop();
            for (auto* v : vfos) { v->stop(); }
            for (auto* k : sinks) { k->stop(); }
            printf("bench %s: %ld blocks of %d in %.4f s = %.1f Msamples/s in (x %d channels out), %.1f us per block\n", kind.c_str(), b1 - b0, bs,
                   sec, (double)(b1 - b0) * bs / sec / 1e6, n, sec / (double)(b1 - b0) * 1e6);
            for (auto* k : sinks) { delete k; }
            for (auto* v : vfos) { delete v; }
            for (auto* l : legs) { delete l; }
            return 0;
        } else if (kind == "hostfir" || kind == "hostvfo") {
            struct Blk { int n; } blk{bs};
            HandlerSource<complex_t> hsrc([](complex_t* d, void* c) { d[0] = complex_t{1.0f, 0.0f}; return static_cast<Blk*>(c)->n; }, &blk);
            struct Flat : filter_window::generic_window {
                int getTapCount() override { return 256; }
                void createTaps(float* t, int n, float f = 1.0f) override { for (int i = 0; i < n; i++) { t[i] = f / (float)n; } }
            } flat;
            FIR<complex_t>* hfir = kind == "hostfir" ? new FIR<complex_t>(&hsrc.out, &flat) : nullptr;
            VFO* hvfo = kind == "hostvfo" ? new VFO(&hsrc.out, inSR * 0.1f, inSR, outSR, outSR) : nullptr;
            HandlerSink<complex_t> hsink(hfir ? &hfir->out : hvfo->out, Count::push, &cnt);
            hsink.start();
            if (hfir) { hfir->start(); } else { hvfo->start(); }
            hsrc.start();
            auto wait_n = [&](long n) {
                const auto t0 = std::chrono::steady_clock::now();
                while (cnt.blocks.load() < n) {
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { return false; }
                }
                return true;
            };
            if (!wait_n(nb / 10 + 2)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
            const long b0 = cnt.blocks.load();
            const auto t0 = std::chrono::steady_clock::now();
            if (!wait_n(b0 + nb)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
            const long b1 = cnt.blocks.load();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            hsrc.stop();
            if (hfir) { hfir->stop(); } else { hvfo->stop(); }
            hsink.stop();
            printf("bench %s: %ld blocks of %d in %.4f s = %.1f Msamples/s in, %.1f us per block\n", kind.c_str(), b1 - b0, bs, sec,
                   (double)(b1 - b0) * bs / sec / 1e6, sec / (double)(b1 - b0) * 1e6);
            delete hfir;
            delete hvfo;
            return 0;
        } else { fprintf(stderr, "bench kind: vfo | chain | split<n> | hostfir | hostvfo\n"); return 2; }
        HandlerSink<complex_t> sink(tail, Count::push, &cnt);
        sink.start();
        if (vfo) { vfo->start(); }
        if (rs) { rs->start(); xl->start(); }
        src.start();
        const int warm = nb / 10 + 2;
        auto wait_for = [&](long n) {
            const auto t0 = std::chrono::steady_clock::now();
            while (cnt.blocks.load() < n) {
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { return false; }
            }
            return true;
        };
        if (!wait_for(warm)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
        const long b0 = cnt.blocks.load();
        const auto t0 = std::chrono::steady_clock::now();
        if (!wait_for(b0 + nb)) { fprintf(stderr, "bench graph timed out\n"); return 3; }
        const long b1 = cnt.blocks.load();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        src.stop();
        if (vfo) { vfo->stop(); }
        if (rs) { xl->stop(); rs->stop(); }
        sink.stop();
        printf("bench %s: %ld blocks of %d in %.4f s = %.1f Msamples/s in, %.1f us per block\n", kind.c_str(), b1 - b0, bs, sec,
               (double)(b1 - b0) * bs / sec / 1e6, sec / (double)(b1 - b0) * 1e6);
        delete vfo;
        delete rs;
        delete xl;
        return 0;
    }
    if (argc < 5) { fprintf(stderr, "missing arguments\n"); return 2; }
    const char* in = argv[2];
    const char* out = argv[3];
    const int block = atoi(argv[4]);
    if (block <= 0 || block > STREAM_BUFFER_SIZE) { fprintf(stderr, "bad block size\n"); return 2; }

    if (mode == "fir" && argc >= 6) {
        FileTaps taps(argv[5]);
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new FIR<complex_t>(s, &taps); });
    }
    if (mode == "fir63") {
        filter_window::BlackmanWindow win(0.1f, 4.0f / 63.0f, 1.0f);
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new FIR<complex_t>(s, &win); });
    }
    if (mode == "firf" && argc >= 6) {
        FileTaps taps(argv[5]);
        return runGraph<float>(in, out, block, [&](stream<float>* s) { return new FIR<float>(s, &taps); });
    }
    if (mode == "firrrc" && argc >= 9) {
        RRCTaps rrc(atoi(argv[5]), (float)atof(argv[6]), (float)atof(argv[7]), (float)atof(argv[8]));
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new FIR<complex_t>(s, &rrc); });
    }
    if (mode == "firbp" && argc >= 9) {
        filter_window::BlackmanBandpassWindow bp((float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7]), (float)atof(argv[8]));
        return runGraph<float>(in, out, block, [&](stream<float>* s) { return new FIR<float>(s, &bp); });
    }
    if (mode == "resampst" && argc >= 9) {
        filter_window::BlackmanWindow win((float)atof(argv[7]), (float)atof(argv[8]), (float)atof(argv[5]));
        const float inSR = (float)atof(argv[5]), outSR = (float)atof(argv[6]);
        return runGraph<stereo_t>(in, out, block, [&](stream<stereo_t>* s) { return new PolyphaseResampler<stereo_t>(s, &win, inSR, outSR); });
    }
    if (mode == "resamp" && argc >= 9) {
        filter_window::BlackmanWindow win((float)atof(argv[7]), (float)atof(argv[8]), (float)atof(argv[5]));
        const float inSR = (float)atof(argv[5]), outSR = (float)atof(argv[6]);
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new PolyphaseResampler<complex_t>(s, &win, inSR, outSR); });
    }
    if (mode == "xlate" && argc >= 7) {
        const float sr = (float)atof(argv[5]), f = (float)atof(argv[6]);
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new FrequencyXlator<complex_t>(s, sr, f); });
    }
    if (mode == "vfo" && argc >= 9) {
        // VFO exposes `stream<complex_t>* out`; adapt it to the harness' blk->out shape
        struct VfoBox {
            VFO v;
            stream<complex_t>& out;
            VfoBox(stream<complex_t>* s, float off, float inSR, float outSR, float bw) : v(s, off, inSR, outSR, bw), out(*v.out) {}
            void start() { v.start(); }
            void stop() { v.stop(); }
        };
        const float off = (float)atof(argv[5]), inSR = (float)atof(argv[6]), outSR = (float)atof(argv[7]), bw = (float)atof(argv[8]);
        return runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new VfoBox(s, off, inSR, outSR, bw); });
    }
    if (mode == "chain" && argc >= 10) {
        // three HIP-backed blocks back to back; only the first reads and the last writes host memory
        FileTaps taps(argv[5]);
        const float sr = (float)atof(argv[6]), f = (float)atof(argv[7]), inSR = (float)atof(argv[8]), outSR = (float)atof(argv[9]);
        struct Chain {
            FrequencyXlator<complex_t> xl;
            FIR<complex_t> fir;
            filter_window::BlackmanWindow win;
            PolyphaseResampler<complex_t> rs;
            stream<complex_t>& out;
            Chain(stream<complex_t>* s, FileTaps* t, float sr, float f, float inSR, float outSR)
                : xl(s, sr, f), fir(&xl.out, t), win(outSR / 2.0f, outSR / 2.0f, inSR), rs(&fir.out, &win, inSR, outSR), out(rs.out) {}
            void start() { rs.start(); fir.start(); xl.start(); }
            void stop() { xl.stop(); fir.stop(); rs.stop(); }
        };
        int rc = runGraph<complex_t>(in, out, block, [&](stream<complex_t>* s) { return new Chain(s, &taps, sr, f, inSR, outSR); });
        return rc;
    }
    if (mode == "math" && argc >= 8) {
        // source -> Splitter -> { FrequencyXlator, identity } -> Add | Substract | Multiply -> sink:
        // out = op(x * nco, x), both inputs of the math block arrive over (device-resident) links
        const std::string opname = argv[5];
        const float sr = (float)atof(argv[6]), f = (float)atof(argv[7]);
        Feed<complex_t> feed;
        feed.data = readAll<complex_t>(in);
        feed.block = block;
        const long nblocks = (long)((feed.data.size() + block - 1) / block);
        HandlerSource<complex_t> src(Feed<complex_t>::pull, &feed);
        Splitter<complex_t> split(&src.out);
        stream<complex_t> la, lb;
        split.bindStream(&la);
        split.bindStream(&lb);
        FrequencyXlator<complex_t> xl(&la, sr, f);
        Collect<complex_t> col;
        int rc = 0;
        auto runWith = [&](auto& blk) {
            HandlerSink<complex_t> sink(&blk.out, Collect<complex_t>::push, &col);
            sink.start();
            blk.start();
            xl.start();
            split.start();
            src.start();
            const auto t0 = std::chrono::steady_clock::now();
            while (col.blocks.load() < nblocks) {
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "math graph timed out\n"); rc = 3; break; }
            }
            src.stop();
            split.stop();
            xl.stop();
            blk.stop();
            sink.stop();
        };
        if (opname == "add") { Add<complex_t> blk(&xl.out, &lb); runWith(blk); }
        else if (opname == "sub") { Substract<complex_t> blk(&xl.out, &lb); runWith(blk); }
        else if (opname == "mul") { Multiply<complex_t> blk(&xl.out, &lb); runWith(blk); }
        else { fprintf(stderr, "math op must be add|sub|mul\n"); return 2; }
        if (rc) { return rc; }
        std::ofstream o(out, std::ios::binary);
        o.write(reinterpret_cast<const char*>(col.data.data()), (std::streamsize)(col.data.size() * sizeof(complex_t)));
        printf("math %s ok: %zu in, %zu out\n", opname.c_str(), feed.data.size(), col.data.size());
        return 0;
    }
    if (mode == "split" && argc >= 9) {
        const int n = atoi(argv[5]);
        const float inSR = (float)atof(argv[6]), outSR = (float)atof(argv[7]), bw = (float)atof(argv[8]);
        Feed<complex_t> feed;
        feed.data = readAll<complex_t>(in);
        feed.block = block;
        const long nblocks = (long)((feed.data.size() + block - 1) / block);
        HandlerSource<complex_t> src(Feed<complex_t>::pull, &feed);
        Splitter<complex_t> split(&src.out);
        std::vector<stream<complex_t>*> links;
        std::vector<VFO*> vfos;
        std::vector<Collect<complex_t>*> cols;
        std::vector<HandlerSink<complex_t>*> sinks;
        for (int i = 0; i < n; i++) {
            links.push_back(new stream<complex_t>());
            const float off = ((float)i - (float)(n - 1) / 2.0f) * inSR / (float)n;
            vfos.push_back(new VFO(links[i], off, inSR, outSR, bw));
            split.bindStream(links[i]);
            cols.push_back(new Collect<complex_t>());
            sinks.push_back(new HandlerSink<complex_t>(vfos[i]->out, Collect<complex_t>::push, cols[i]));
        }
        for (auto* s : sinks) { s->start(); }
        for (auto* v : vfos) { v->start(); }
        split.start();
        src.start();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            while (cols[i]->blocks.load() < nblocks) {
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "split graph timed out\n"); return 3; }
            }
        }
        src.stop();
        split.stop();
        for (auto* v : vfos) { v->stop(); }
        for (auto* s : sinks) { s->stop(); }
        for (int i = 0; i < n; i++) {
            std::ofstream o(std::string(out) + "." + std::to_string(i) + ".cf32", std::ios::binary);
            o.write(reinterpret_cast<const char*>(cols[i]->data.data()), (std::streamsize)(cols[i]->data.size() * sizeof(complex_t)));
        }
        printf("split ok: %d channels, %zu in, %zu out each\n", n, feed.data.size(), cols[0]->data.size());
        for (auto* s : sinks) { delete s; }
        for (auto* v : vfos) { delete v; }
        for (auto* l : links) { delete l; }
        for (auto* c : cols) { delete c; }
        return 0;
    }
    if (mode == "splitretune" && argc >= 11) {
        const int n = atoi(argv[5]);
        const float inSR = (float)atof(argv[6]), outSR = (float)atof(argv[7]), bw = (float)atof(argv[8]);
        const long K = atol(argv[9]);
        const float newOff = (float)atof(argv[10]);
        const bool reconf = argc >= 12 && std::string(argv[11]) == "reconf";
        // "rebind": an extra leg (link + VFO + sink, created up front) is bound to the LIVE Splitter right before block K is fed
        // and unbound again before block K + 3 -- without waiting for the graph to go idle, so tokens of the banked blocks
        // K - 1 / K + 2 may still sit in the links when the bank is taken down (they must still be read as tokens)
        const bool rebind = argc >= 12 && std::string(argv[11]) == "rebind";
        struct Gate {
            Feed<complex_t> feed;
            std::vector<Collect<complex_t>*>* cols = nullptr;
            std::vector<VFO*>* vfos = nullptr;
            long K = 0, fed = 0;
            float newOff = 0, bw2 = 0;
            bool reconf = false, rebind = false;
            Splitter<complex_t>* split = nullptr;
            stream<complex_t>* extra = nullptr;
            stream<complex_t>* srcOut = nullptr;
            static int pull(complex_t* dst, void* ctx) {
                Gate* g = static_cast<Gate*>(ctx);
                if (g->rebind) {
                    if (g->fed == g->K || g->fed == g->K + 3) {
                        // the Splitter has taken the previous block (it flushed its input: tokens are in the links, the worker is back
                        // in read()) -- a stop in the middle of a block would drop that block, as in the reference -- but the VFOs
                        // have not necessarily read their tokens yet: that is the window the re-plumbing must survive
                        (void)g->srcOut->waitFlushed();
                        if (g->fed == g->K) { g->split->bindStream(g->extra); }
                        else { g->split->unbindStream(g->extra); }
                    }
                } else if (g->fed == g->K) {
                    // every block fed so far has come out of every sink: the graph is idle, the change lands exactly here
                    for (auto* c : *g->cols) {
                        while (c->blocks.load() < g->K) { std::this_thread::sleep_for(std::chrono::microseconds(200)); }
                    }
                    if (g->reconf) { (*g->vfos)[0]->setBandwidth(g->bw2); }
                    else { (*g->vfos)[1]->setOffset(g->newOff); }
                }
                g->fed++;
                return Feed<complex_t>::pull(dst, &g->feed);
            }
        };
        Gate gate;
        gate.feed.data = readAll<complex_t>(in);
        gate.feed.block = block;
        gate.K = K;
        gate.newOff = newOff;
        gate.bw2 = bw * 0.5f;
        gate.reconf = reconf;
        gate.rebind = rebind;
        const long nblocks = (long)((gate.feed.data.size() + block - 1) / block);
        HandlerSource<complex_t> src(Gate::pull, &gate);
        Splitter<complex_t> split(&src.out);
        std::vector<stream<complex_t>*> links;
        std::vector<VFO*> vfos;
        std::vector<Collect<complex_t>*> cols;
        std::vector<HandlerSink<complex_t>*> sinks;
        for (int i = 0; i < n + (rebind ? 1 : 0); i++) {
            links.push_back(new stream<complex_t>());
            const float off = i < n ? ((float)i - (float)(n - 1) / 2.0f) * inSR / (float)n : 0.125f * inSR;
            vfos.push_back(new VFO(links[i], off, inSR, outSR, bw));
            if (i < n) { split.bindStream(links[i]); }
            cols.push_back(new Collect<complex_t>());
            sinks.push_back(new HandlerSink<complex_t>(vfos[i]->out, Collect<complex_t>::push, cols[i]));
        }
        gate.cols = &cols;
        gate.vfos = &vfos;
        gate.split = &split;
        gate.extra = rebind ? links[n] : nullptr;
        gate.srcOut = &src.out;
        for (auto* s : sinks) { s->start(); }
        for (auto* v : vfos) { v->start(); }
        split.start();
        src.start();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n + (rebind ? 1 : 0); i++) {
            while (cols[i]->blocks.load() < (i < n ? nblocks : 3)) {
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "splitretune graph timed out\n"); return 3; }
            }
        }
        src.stop();
        split.stop();
        for (auto* v : vfos) { v->stop(); }
        for (auto* s : sinks) { s->stop(); }
        for (int i = 0; i < n + (rebind ? 1 : 0); i++) {
            std::ofstream o(std::string(out) + "." + std::to_string(i) + ".cf32", std::ios::binary);
            o.write(reinterpret_cast<const char*>(cols[i]->data.data()), (std::streamsize)(cols[i]->data.size() * sizeof(complex_t)));
        }
        printf("splitretune ok: %d channels, %zu in, %zu out each\n", n, gate.feed.data.size(), cols[0]->data.size());
        for (auto* s : sinks) { delete s; }
        for (auto* v : vfos) { delete v; }
        for (auto* l : links) { delete l; }
        for (auto* c : cols) { delete c; }
        return 0;
    }
    if (mode == "mulsplit" && argc >= 9) {
        const int n = atoi(argv[5]);
        const float inSR = (float)atof(argv[6]), outSR = (float)atof(argv[7]), bw = (float)atof(argv[8]);
        Feed<complex_t> feed;
        feed.data = readAll<complex_t>(in);
        feed.block = block;
        const long nblocks = (long)((feed.data.size() + block - 1) / block);
        HandlerSource<complex_t> src(Feed<complex_t>::pull, &feed);
        Splitter<complex_t> split0(&src.out);
        stream<complex_t> la, lb;
        split0.bindStream(&la);
        split0.bindStream(&lb);
        Multiply<complex_t> mul(&la, &lb);
        Splitter<complex_t> split(&mul.out);
        std::vector<stream<complex_t>*> links;
        std::vector<VFO*> vfos;
        std::vector<Collect<complex_t>*> cols;
        std::vector<HandlerSink<complex_t>*> sinks;
        for (int i = 0; i < n; i++) {
            links.push_back(new stream<complex_t>());
            const float off = ((float)i - (float)(n - 1) / 2.0f) * inSR / (float)n;
            vfos.push_back(new VFO(links[i], off, inSR, outSR, bw));
            split.bindStream(links[i]);
            cols.push_back(new Collect<complex_t>());
            sinks.push_back(new HandlerSink<complex_t>(vfos[i]->out, Collect<complex_t>::push, cols[i]));
        }
        for (auto* s : sinks) { s->start(); }
        for (auto* v : vfos) { v->start(); }
        split.start();
        mul.start();
        split0.start();
        src.start();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            while (cols[i]->blocks.load() < nblocks) {
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "mulsplit graph timed out\n"); return 3; }
            }
        }
        src.stop();
        split0.stop();
        mul.stop();
        split.stop();
        for (auto* v : vfos) { v->stop(); }
        for (auto* s : sinks) { s->stop(); }
        for (int i = 0; i < n; i++) {
            std::ofstream o(std::string(out) + "." + std::to_string(i) + ".cf32", std::ios::binary);
            o.write(reinterpret_cast<const char*>(cols[i]->data.data()), (std::streamsize)(cols[i]->data.size() * sizeof(complex_t)));
        }
        printf("mulsplit ok: %d channels, %zu in, %zu out each\n", n, feed.data.size(), cols[0]->data.size());
        for (auto* s : sinks) { delete s; }
        for (auto* v : vfos) { delete v; }
        for (auto* l : links) { delete l; }
        for (auto* c : cols) { delete c; }
        return 0;
    }
    if (mode == "wavfir") {
        // BASELINE config 1: int16 stereo (I,Q) WAV -> complex (s / 32768.0f) -> 63-tap FIR.
        WavReader rd(in);
        if (!rd.isValid() || rd.getBitDepth() != 16 || rd.getChannelCount() != 2) { fprintf(stderr, "need a 16-bit 2-channel WAV\n"); return 2; }
        const size_t frames = rd.hdr.dataSize / 4;
        std::vector<int16_t> pcm(frames * 2);
        rd.readSamples(pcm.data(), pcm.size() * 2);
        std::vector<complex_t> iq(frames);
        for (size_t i = 0; i < frames; i++) { iq[i] = {pcm[2 * i] / 32768.0f, pcm[2 * i + 1] / 32768.0f}; }
        const std::string tmp = std::string(out) + ".in.cf32";
        { std::ofstream o(tmp, std::ios::binary); o.write(reinterpret_cast<const char*>(iq.data()), (std::streamsize)(iq.size() * 8)); }
        filter_window::BlackmanWindow win(0.1f * rd.getSampleRate(), 4.0f * rd.getSampleRate() / 63.0f, (float)rd.getSampleRate());
        return runGraph<complex_t>(tmp.c_str(), out, block, [&](stream<complex_t>* s) { return new FIR<complex_t>(s, &win); });
    }
    if (mode == "shard" && argc >= 6) {
        const std::vector<complex_t> x = readAll<complex_t>(in);
        const std::vector<float> taps = readAll<float>(argv[5]);
        const int rank = argc >= 9 ? atoi(argv[6]) : 0, world = argc >= 9 ? atoi(argv[7]) : 1;
        const int n = block, H = (int)taps.size() - 1;
        const long steps = (long)(x.size() / ((size_t)n * world));
        if (steps < 1 || H < 1 || n < H) { fprintf(stderr, "shard: need at least one chunk of >= ntaps - 1 samples per rank\n"); return 2; }
        unsigned char id[QDSP_HIP_RING_ID_BYTES];
        if (rank == 0) {
            if (qdsp_hip_ring_unique_id(id) != 0) { fprintf(stderr, "shard: no RCCL\n"); return 3; }
            if (argc >= 9) { std::ofstream o(std::string(argv[8]) + ".tmp", std::ios::binary); o.write((const char*)id, sizeof(id)); o.close(); rename((std::string(argv[8]) + ".tmp").c_str(), argv[8]); }
        } else {
            for (int tries = 0;; tries++) {
                std::ifstream f(argv[8], std::ios::binary);
                if (f && f.read((char*)id, sizeof(id))) { break; }
                if (tries > 3000) { fprintf(stderr, "shard: no id file\n"); return 3; }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        const int dev = detail::hipDeviceForBlocks();
        void *ring = nullptr, *fir = nullptr, *d_in[2] = {nullptr, nullptr}, *d_out = nullptr;
        int rc = qdsp_hip_ring_create(&ring, dev, rank, world, id, H * (int)sizeof(complex_t));
        if (rc == 0) { rc = qdsp_hip_fir_cf32_create(&fir, dev, taps.data(), (int)taps.size(), 0); }
        for (int i = 0; i < 2 && rc == 0; i++) { rc = qdsp_hip_dev_alloc(dev, &d_in[i], (size_t)n * sizeof(complex_t)); }
        if (rc == 0) { rc = qdsp_hip_dev_alloc(dev, &d_out, (size_t)n * sizeof(complex_t)); }
        if (rc != 0) { fprintf(stderr, "shard: setup failed: %s\n", qdsp_hip_error_string(rc)); return 3; }
        auto chunk = [&](long s) { return x.data() + ((size_t)s * world + rank) * n; };
        std::vector<complex_t> y((size_t)steps * n);
        // step s: [upload s + 1] -> complete(s) -> history -> post(s + 1) -> kernel(s): the exchange of the next step runs under this kernel
        rc = qdsp_hip_memcpy_h2d(dev, d_in[0], chunk(0), (size_t)n * sizeof(complex_t));
        if (rc == 0) { rc = qdsp_hip_ring_post(ring, static_cast<complex_t*>(d_in[0]) + (n - H), nullptr); }
        for (long s = 0; s < steps && rc == 0; s++) {
            void* cur = d_in[s & 1];
            void* nxt = d_in[(s + 1) & 1];
            if (s + 1 < steps) { rc = qdsp_hip_memcpy_h2d(dev, nxt, chunk(s + 1), (size_t)n * sizeof(complex_t)); }
            const void *halo = nullptr, *prev = nullptr;
            if (rc == 0) { rc = qdsp_hip_ring_complete(ring, nullptr, &halo, &prev); }
            // ranks > 0: the predecessor's tail of THIS step; rank 0: the last rank's tail of the step before (zeros at the start)
            if (rc == 0) { rc = qdsp_hip_fir_cf32_set_history_dev(fir, rank == 0 ? prev : halo, nullptr); }
            if (rc == 0 && s + 1 < steps) { rc = qdsp_hip_ring_post(ring, static_cast<complex_t*>(nxt) + (n - H), nullptr); }
            if (rc == 0) { const long long r2 = qdsp_hip_fir_cf32_process_dev(fir, cur, n, d_out, nullptr); rc = r2 < 0 ? (int)r2 : 0; }
            if (rc == 0) { rc = qdsp_hip_memcpy_d2h(dev, y.data() + (size_t)s * n, d_out, (size_t)n * sizeof(complex_t)); }
        }
        if (rc != 0) { fprintf(stderr, "shard: %s\n", qdsp_hip_error_string(rc)); return 3; }
        (void)qdsp_hip_ring_drain(ring);
        std::ofstream o(world > 1 ? std::string(out) + "." + std::to_string(rank) : std::string(out), std::ios::binary);
        o.write(reinterpret_cast<const char*>(y.data()), (std::streamsize)(y.size() * sizeof(complex_t)));
        printf("shard ok: rank %d of %d, %ld chunks of %d samples, halo %d samples over the ring\n", rank, world, steps, n, H);
        qdsp_hip_fir_cf32_destroy(fir);
        qdsp_hip_ring_destroy(ring);
        for (void* p : d_in) { qdsp_hip_dev_free(dev, p); }
        qdsp_hip_dev_free(dev, d_out);
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}
