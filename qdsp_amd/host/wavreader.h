// wavreader.h -- WavReader (public surface and behaviour of the reference's src/wavreader.h:14-84):
// checks the "RIFF"/"WAVE" tags, exposes bit depth / channel count / sample rate, and
// readSamples() WRAPS AROUND to the first frame when the file runs out (wavreader.h:41-50), so a
// short capture loops forever.  The header is parsed byte-wise by wav_detail::Header (wav.h).
#pragma once
#include "wav.h"

class WavReader {
public:
    WavReader(std::string path) : src(path.c_str(), std::ios::binary) {
        unsigned char raw[wav_detail::kHeaderBytes] = {};
        src.read(reinterpret_cast<char*>(raw), (std::streamsize)sizeof(raw));
        if (src.gcount() == (std::streamsize)sizeof(raw)) { hdr.get(raw); }
    }

    bool isValid() { return hdr.tagsOk; }
    uint32_t getSampleRate() { return hdr.sampleRate; }
    uint16_t getChannelCount() { return hdr.channelCount; }
    uint16_t getBitDepth() { return hdr.bitDepth; }

    // fills `data` completely; at end of file continues from the first frame
    void readSamples(void* data, size_t size) {
        char* dst = static_cast<char*>(data);
        size_t have = pull(dst, size);
        if (have < size) {
            rewind();
            have += pull(dst + have, size - have);
        }
        consumed += size;
    }

    void rewind() {
        src.clear();
        src.seekg((std::streamoff)wav_detail::kHeaderBytes);
    }

    void close() { src.close(); }

    wav_detail::Header hdr;      // parsed fields (dataSize = payload bytes the header announces)

private:
    size_t pull(char* dst, size_t want) {
        src.read(dst, (std::streamsize)want);
        return (size_t)src.gcount();
    }

    std::ifstream src;
    size_t consumed = 0;
};
