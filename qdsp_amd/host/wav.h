// wav.h -- WavWriter: canonical 44-byte RIFF/WAVE PCM header + raw sample data
// (on-disk format and API of the reference's src/wav.h:11-62).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>

#define WAV_SIGNATURE "RIFF"
#define WAV_TYPE "WAVE"
#define WAV_FORMAT_MARK "fmt "
#define WAV_DATA_MARK "data"
#define WAV_SAMPLE_TYPE_PCM 1

// The one header layout both reader and writer use (packed by construction: 44 bytes).
struct WavHeader_t {
    char signature[4];            // "RIFF"
    uint32_t fileSize;            // data bytes + sizeof(WavHeader_t) - 8
    char fileType[4];             // "WAVE"
    char formatMarker[4];         // "fmt "
    uint32_t formatHeaderLength;  // 16
    uint16_t sampleType;          // PCM = 1
    uint16_t channelCount;
    uint32_t sampleRate;
    uint32_t bytesPerSecond;
    uint16_t bytesPerSample;      // per frame (all channels)
    uint16_t bitDepth;
    char dataMarker[4];           // "data"
    uint32_t dataSize;
};
static_assert(sizeof(WavHeader_t) == 44, "canonical WAV header is 44 bytes");

class WavWriter {
public:
    WavWriter(std::string path, uint16_t bitDepth, uint16_t channelCount, uint32_t sampleRate) {
        file = std::ofstream(path.c_str(), std::ios::binary);
        std::memset(&hdr, 0, sizeof(hdr));
        std::memcpy(hdr.signature, WAV_SIGNATURE, 4);
        std::memcpy(hdr.fileType, WAV_TYPE, 4);
        std::memcpy(hdr.formatMarker, WAV_FORMAT_MARK, 4);
        std::memcpy(hdr.dataMarker, WAV_DATA_MARK, 4);
        hdr.formatHeaderLength = 16;
        hdr.sampleType = WAV_SAMPLE_TYPE_PCM;
        hdr.channelCount = channelCount;
        hdr.sampleRate = sampleRate;
        hdr.bytesPerSecond = (bitDepth / 8) * channelCount * sampleRate;
        hdr.bytesPerSample = (bitDepth / 8) * channelCount;
        hdr.bitDepth = bitDepth;
        file.write(reinterpret_cast<const char*>(&hdr), sizeof(hdr));
    }

    bool isOpen() { return file.is_open(); }

    void writeSamples(void* data, size_t size) {
        file.write(static_cast<const char*>(data), (std::streamsize)size);
        bytesWritten += size;
    }

    // patches the two size fields, as the reference does on close()
    void close() {
        hdr.fileSize = (uint32_t)(bytesWritten + sizeof(WavHeader_t) - 8);
        hdr.dataSize = (uint32_t)bytesWritten;
        file.seekp(0);
        file.write(reinterpret_cast<const char*>(&hdr), sizeof(hdr));
        file.close();
    }

private:
    std::ofstream file;
    size_t bytesWritten = 0;
    WavHeader_t hdr;
};
