// patch_b_fir -- INTEGRATION.md section B, compiled: what FIR<complex_t>::run() looks like in a tree that
// keeps its own src/dsp headers and only swaps the VOLK loop for one call into libqdsp_hip.so.  The
// stream/window stand-ins below carry just the members the patched lines touch (writeBuf / readBuf /
// read / flush / swap, getTapCount / createTaps); everything between the "patched" markers is the text
// INTEGRATION.md shows.  Usage: patch_b_fir <in.cf32> <out.cf32> <block> <taps.f32>
#include <qdsp_hip.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#define STREAM_BUFFER_SIZE 1000000

struct complex_t { float re, im; };

template <class T> struct mini_stream {       // the protocol of src/dsp/stream.h:21-125, single-threaded here
    T* writeBuf = nullptr;
    T* readBuf = nullptr;
    int pending = -1;
    mini_stream() {
        void* a = nullptr; void* b = nullptr;
        qdsp_hip_host_alloc(&a, STREAM_BUFFER_SIZE * sizeof(T));   // volk_malloc in the reference; pinned here
        qdsp_hip_host_alloc(&b, STREAM_BUFFER_SIZE * sizeof(T));
        writeBuf = static_cast<T*>(a);
        readBuf = static_cast<T*>(b);
    }
    ~mini_stream() { qdsp_hip_host_free(writeBuf); qdsp_hip_host_free(readBuf); }
    bool swap(int n) { T* t = writeBuf; writeBuf = readBuf; readBuf = t; pending = n; return true; }
    int read() { return pending; }
    void flush() { pending = -1; }
};

struct file_taps {                              // a generic_window (src/dsp/window.h:7-11)
    std::vector<float> t;
    int getTapCount() { return (int)t.size(); }
    void createTaps(float* taps, int n, float factor = 1.0f) { for (int i = 0; i < n; i++) taps[i] = t[i] * factor; }
};

struct PatchedFIR {
    mini_stream<complex_t>* _in = nullptr;
    mini_stream<complex_t> out;
    float* taps = nullptr;
    int tapCount = 0;
    void* hip = nullptr;

    int init(mini_stream<complex_t>* in, file_taps* window) {
        _in = in;
        tapCount = window->getTapCount();
        taps = static_cast<float*>(malloc(tapCount * sizeof(float)));
        window->createTaps(taps, tapCount);
        // ---- patched (filter.h:24-26) -------------------------------------------------------------
        return qdsp_hip_fir_cf32_create(&hip, /*device*/ 0, taps, tapCount, STREAM_BUFFER_SIZE);
        // ---------------------------------------------------------------------------------------------
    }
    int run() {
        // ---- patched: replaces filter.h:55-71 (memcpy, the volk_32fc_32f_dot_prod_32fc loop, memmove) ------
        int count = _in->read();
        if (count < 0) { return -1; }
        int rc = qdsp_hip_fir_cf32_process(hip, (const float*)_in->readBuf, count, (float*)out.writeBuf);
        _in->flush();
        if (rc != 0 || !out.swap(count)) { return -1; }
        return count;
        // ---------------------------------------------------------------------------------------------
    }
    ~PatchedFIR() { qdsp_hip_fir_cf32_destroy(hip); free(taps); }
};

template <class T> static std::vector<T> readAll(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<T> v;
    if (!f) { return v; }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    v.resize((size_t)n / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
    return v;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: patch_b_fir <in.cf32> <out.cf32> <block> <taps.f32>\n"); return 2; }
    const std::vector<complex_t> x = readAll<complex_t>(argv[1]);
    const int block = atoi(argv[3]);
    file_taps win;
    win.t = readAll<float>(argv[4]);
    if (x.empty() || win.t.empty() || block <= 0 || block > STREAM_BUFFER_SIZE) { fprintf(stderr, "bad arguments\n"); return 2; }
    mini_stream<complex_t> src;
    PatchedFIR fir;
    const int rc = fir.init(&src, &win);
    if (rc != 0) { fprintf(stderr, "init: %s\n", qdsp_hip_error_string(rc)); return 1; }
    std::ofstream o(argv[2], std::ios::binary);
    for (size_t pos = 0; pos < x.size(); pos += (size_t)block) {
        const int n = (int)std::min<size_t>((size_t)block, x.size() - pos);
        for (int i = 0; i < n; i++) { src.writeBuf[i] = x[pos + i]; }
        src.swap(n);
        const int got = fir.run();
        if (got != n) { fprintf(stderr, "run() returned %d\n", got); return 1; }
        o.write(reinterpret_cast<const char*>(fir.out.readBuf), (std::streamsize)((size_t)got * sizeof(complex_t)));
        fir.out.flush();
    }
    printf("patch_b_fir ok: %zu samples\n", x.size());
    return 0;
}
