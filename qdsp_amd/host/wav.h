// wav.h -- WavWriter for canonical PCM WAVE files: a 44-byte RIFF header followed by raw frames
// (same on-disk format and public surface as the reference's src/wav.h:11-62: constructor
// (path, bitDepth, channelCount, sampleRate), isOpen, writeSamples, close).
//
// The header is never mapped onto a struct here: it is written and parsed field by field as
// little-endian bytes (wav_detail::Header::put / get), so the code does not depend on struct
// packing or host endianness.  Layout (offset: field):
//    0 "RIFF"   4 riff size = 36 + data bytes   8 "WAVE"   12 "fmt "   16 fmt length = 16
//   20 format tag (1 = PCM)   22 channels   24 frames per second   28 bytes per second
//   32 bytes per frame   34 bits per sample   36 "data"   40 data bytes
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>

namespace wav_detail {

constexpr size_t kHeaderBytes = 44;

struct Header {
    uint16_t formatTag = 1;
    uint16_t channelCount = 0;
    uint32_t sampleRate = 0;
    uint16_t bitDepth = 0;
    uint32_t dataSize = 0;       // bytes of sample data that follow the header
    bool tagsOk = false;         // get(): "RIFF" and "WAVE" were where they belong

    uint32_t frameBytes() const { return (uint32_t)(bitDepth / 8) * channelCount; }

    static void le16(unsigned char* p, uint16_t v) { p[0] = (unsigned char)(v & 0xff); p[1] = (unsigned char)(v >> 8); }
    static void le32(unsigned char* p, uint32_t v) { for (int i = 0; i < 4; i++) { p[i] = (unsigned char)((v >> (8 * i)) & 0xff); } }
    static uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
    static uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

    void put(unsigned char (&b)[kHeaderBytes]) const {
        std::memcpy(b + 0, "RIFF", 4);
        le32(b + 4, (uint32_t)(kHeaderBytes - 8) + dataSize);
        std::memcpy(b + 8, "WAVE", 4);
        std::memcpy(b + 12, "fmt ", 4);
        le32(b + 16, 16);
        le16(b + 20, formatTag);
        le16(b + 22, channelCount);
        le32(b + 24, sampleRate);
        le32(b + 28, frameBytes() * sampleRate);
        le16(b + 32, (uint16_t)frameBytes());
        le16(b + 34, bitDepth);
        std::memcpy(b + 36, "data", 4);
        le32(b + 40, dataSize);
    }

    void get(const unsigned char (&b)[kHeaderBytes]) {
        tagsOk = std::memcmp(b + 0, "RIFF", 4) == 0 && std::memcmp(b + 8, "WAVE", 4) == 0;
        formatTag = rd16(b + 20);
        channelCount = rd16(b + 22);
        sampleRate = rd32(b + 24);
        bitDepth = rd16(b + 34);
        dataSize = rd32(b + 40);
    }
};

}  // namespace wav_detail

class WavWriter {
public:
    WavWriter(std::string path, uint16_t bitDepth, uint16_t channelCount, uint32_t sampleRate) : out(path.c_str(), std::ios::binary) {
        head.bitDepth = bitDepth;
        head.channelCount = channelCount;
        head.sampleRate = sampleRate;
        emitHeader();                       // sizes are patched in by close()
    }

    bool isOpen() { return out.is_open(); }

    void writeSamples(void* data, size_t size) {
        out.write(static_cast<const char*>(data), (std::streamsize)size);
        payload += size;
    }

    // rewrites the header with the final sizes (as the reference does) and closes the file
    void close() {
        head.dataSize = (uint32_t)payload;
        out.seekp(0);
        emitHeader();
        out.close();
    }

private:
    void emitHeader() {
        unsigned char raw[wav_detail::kHeaderBytes];
        head.put(raw);
        out.write(reinterpret_cast<const char*>(raw), (std::streamsize)sizeof(raw));
    }

    std::ofstream out;
    wav_detail::Header head;
    size_t payload = 0;
};
