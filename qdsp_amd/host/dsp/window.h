// dsp/window.h -- tap designers.  They run once, on the host, at init()/updateWindow();
// the taps they produce are what is uploaded to the GPU, so they must equal the reference's
// bit for bit.  The formulas below are therefore the reference's, evaluated in the same
// float operations in the same order (src/dsp/window.h:36-70 BlackmanWindow, :105-140
// BlackmanBandpassWindow, :161-225 RRCTaps) -- including its quirks, which are NOT fixed:
//   * the "Blackman" factor has no tap index in it (and 0.8 where a Blackman window has
//     0.08), so it is a constant that cancels in the normalisation: the taps are a plain
//     truncated sinc centred on tapCount/2 (not symmetric for odd counts);
//   * an even tap count puts 0/0 at i == tapCount/2 (NaN): getTapCount() never returns one;
//   * RRCTaps forces the count odd with `|= 1` and so needs taps[] sized for it.
// Anything else (e.g. a 256-tap design) enters through a custom generic_window subclass.
#pragma once
#include <cmath>

#include "types.h"

namespace dsp {
namespace filter_window {

class generic_window {
public:
    virtual ~generic_window() {}
    virtual int getTapCount() { return -1; }
    virtual void createTaps(float* taps, int tapCount, float factor = 1.0f) { (void)taps; (void)tapCount; (void)factor; }
};

namespace detail {
// tap count shared by both Blackman designers: 4 / (transition width in cycles/sample),
// at least 4, bumped to odd.
inline int blackmanTapCount(float transWidth, float sampleRate) {
    int n = (int)(4.0f / (transWidth / sampleRate));
    if (n < 4) { n = 4; }
    return (n % 2 == 0) ? n + 1 : n;
}

// raw (un-normalised) taps into `taps`, returns their float sum accumulated in tap order
inline float blackmanRawTaps(float* taps, int tapCount, float cutoff, float sampleRate) {
    float fc = cutoff / sampleRate;
    if (fc > 1.0f) { fc = 1.0f; }
    const float tc = (float)tapCount;
    float sum = 0.0f;
    for (int i = 0; i < tapCount; i++) {
        const float d = (float)i - (tc / 2);
        const float v = (sinf(2.0f * FL_M_PI * fc * d) / d) *
                        (0.42f - (0.5f * cosf(2.0f * FL_M_PI / tc)) + (0.8f * cosf(4.0f * FL_M_PI / tc)));
        taps[i] = v;
        sum += v;
    }
    return sum;
}
}  // namespace detail

class BlackmanWindow : public generic_window {
public:
    BlackmanWindow() {}
    BlackmanWindow(float cutoff, float transWidth, float sampleRate) { init(cutoff, transWidth, sampleRate); }

    void init(float cutoff, float transWidth, float sampleRate) {
        _cutoff = cutoff;
        _transWidth = transWidth;
        _sampleRate = sampleRate;
    }

    void setSampleRate(float sampleRate) { _sampleRate = sampleRate; }
    void setCutoff(float cutoff) { _cutoff = cutoff; }
    void setTransWidth(float transWidth) { _transWidth = transWidth; }

    int getTapCount() override { return detail::blackmanTapCount(_transWidth, _sampleRate); }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        const float sum = detail::blackmanRawTaps(taps, tapCount, _cutoff, _sampleRate);
        for (int i = 0; i < tapCount; i++) {
            taps[i] *= factor;
            taps[i] /= sum;
        }
    }

private:
    float _cutoff = 0.0f, _transWidth = 1.0f, _sampleRate = 1.0f;
};

class BlackmanBandpassWindow : public generic_window {
public:
    BlackmanBandpassWindow() {}
    BlackmanBandpassWindow(float cutoff, float transWidth, float offset, float sampleRate) { init(cutoff, transWidth, offset, sampleRate); }

    void init(float cutoff, float transWidth, float offset, float sampleRate) {
        _cutoff = cutoff;
        _transWidth = transWidth;
        _offset = offset;
        _sampleRate = sampleRate;
    }

    void setSampleRate(float sampleRate) { _sampleRate = sampleRate; }
    void setCutoff(float cutoff) { _cutoff = cutoff; }
    void setTransWidth(float transWidth) { _transWidth = transWidth; }
    void setOffset(float offset) { _offset = offset; }

    int getTapCount() override { return detail::blackmanTapCount(_transWidth, _sampleRate); }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        const float sum = detail::blackmanRawTaps(taps, tapCount, _cutoff, _sampleRate);
        for (int i = 0; i < tapCount; i++) {
            taps[i] *= cosf(2.0f * (_offset / _sampleRate) * FL_M_PI * (float)i);  // shift to the band centre
            taps[i] *= factor;
            taps[i] /= sum;
        }
    }

private:
    float _cutoff = 0.0f, _transWidth = 1.0f, _sampleRate = 1.0f, _offset = 0.0f;
};

}  // namespace filter_window

// Root-raised-cosine taps (the reference credits GNU Radio's firdes::root_raised_cosine).
// Arithmetic is double with the float FL_M_PI and a float samples-per-symbol ratio, exactly
// as in the reference (src/dsp/window.h:181-225).
class RRCTaps : public filter_window::generic_window {
public:
    RRCTaps() {}
    RRCTaps(int tapCount, float sampleRate, float baudRate, float alpha) { init(tapCount, sampleRate, baudRate, alpha); }

    void init(int tapCount, float sampleRate, float baudRate, float alpha) {
        _tapCount = tapCount;
        _sampleRate = sampleRate;
        _baudRate = baudRate;
        _alpha = alpha;
    }

    int getTapCount() override { return _tapCount; }
    void setSampleRate(float sampleRate) { _sampleRate = sampleRate; }
    void setBaudRate(float baudRate) { _baudRate = baudRate; }
    void setTapCount(int count) { _tapCount = count; }
    void setAlpha(float alpha) { _alpha = alpha; }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        (void)factor;  // the reference ignores it too
        tapCount |= 1;
        const double spb = _sampleRate / _baudRate;  // samples per symbol
        const int mid = tapCount / 2;
        double scale = 0;
        for (int i = 0; i < tapCount; i++) {
            const double xi = i - mid;
            double x1 = FL_M_PI * xi / spb;
            double x2 = 4 * _alpha * xi / spb;
            double x3 = x2 * x2 - 1;
            double num, den;
            if (fabs(x3) >= 0.000001) {
                if (i != mid) {
                    num = cos((1 + _alpha) * x1) + sin((1 - _alpha) * x1) / (4 * _alpha * xi / spb);
                } else {
                    num = cos((1 + _alpha) * x1) + (1 - _alpha) * FL_M_PI / (4 * _alpha);
                }
                den = x3 * FL_M_PI;
            } else {
                if (_alpha == 1) {
                    taps[i] = -1;
                    scale += taps[i];
                    continue;
                }
                x3 = (1 - _alpha) * x1;
                x2 = (1 + _alpha) * x1;
                num = (sin(x2) * (1 + _alpha) * FL_M_PI - cos(x3) * ((1 - _alpha) * FL_M_PI * spb) / (4 * _alpha * xi) +
                       sin(x3) * spb * spb / (4 * _alpha * xi * xi));
                den = -32 * FL_M_PI * _alpha * _alpha * xi / spb;
            }
            taps[i] = 4 * _alpha * num / den;
            scale += taps[i];
        }
        for (int i = 0; i < tapCount; i++) { taps[i] = taps[i] / scale; }
    }

private:
    int _tapCount = 0;
    float _sampleRate = 1.0f, _baudRate = 1.0f, _alpha = 0.35f;
};

}  // namespace dsp
