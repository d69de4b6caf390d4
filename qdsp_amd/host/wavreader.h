// wavreader.h -- WavReader (API and behaviour of the reference's src/wavreader.h:14-84):
// validates "RIFF"/"WAVE", exposes the header fields, and readSamples() WRAPS AROUND to the
// start of the data at end of file (wavreader.h:41-50) so a short capture loops forever.
#pragma once
#include "wav.h"

class WavReader {
public:
    WavReader(std::string path) {
        file = std::ifstream(path.c_str(), std::ios::binary);
        std::memset(&hdr, 0, sizeof(hdr));
        file.read(reinterpret_cast<char*>(&hdr), sizeof(WavHeader_t));
        valid = file.gcount() == (std::streamsize)sizeof(WavHeader_t) && std::memcmp(hdr.signature, "RIFF", 4) == 0 &&
                std::memcmp(hdr.fileType, "WAVE", 4) == 0;
    }

    uint16_t getBitDepth() { return hdr.bitDepth; }
    uint16_t getChannelCount() { return hdr.channelCount; }
    uint32_t getSampleRate() { return hdr.sampleRate; }
    bool isValid() { return valid; }

    void readSamples(void* data, size_t size) {
        char* dst = static_cast<char*>(data);
        file.read(dst, (std::streamsize)size);
        const size_t got = (size_t)file.gcount();
        if (got < size) {
            file.clear();
            file.seekg(sizeof(WavHeader_t));
            file.read(dst + got, (std::streamsize)(size - got));
        }
        bytesRead += size;
    }

    void rewind() {
        file.clear();
        file.seekg(sizeof(WavHeader_t));
    }

    void close() { file.close(); }

    WavHeader_t hdr;

private:
    bool valid = false;
    std::ifstream file;
    size_t bytesRead = 0;
};
