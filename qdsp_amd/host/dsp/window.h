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

// cutoff / transition width / sample rate (+ band centre): the state both Blackman designers keep, with the
// reference's setter names
class blackman_params : public generic_window {
public:
    void setSampleRate(float sampleRate) { _sampleRate = sampleRate; }
    void setCutoff(float cutoff) { _cutoff = cutoff; }
    void setTransWidth(float transWidth) { _transWidth = transWidth; }

    int getTapCount() override { return detail::blackmanTapCount(_transWidth, _sampleRate); }

protected:
    void keep(float cutoff, float transWidth, float sampleRate) {
        _cutoff = cutoff;
        _transWidth = transWidth;
        _sampleRate = sampleRate;
    }
    // taps[i] = raw[i] * shift(i) * factor / sum(raw), each step rounded to float in this order
    template <class SHIFT> void design(float* taps, int tapCount, float factor, SHIFT shift) {
        const float total = detail::blackmanRawTaps(taps, tapCount, _cutoff, _sampleRate);
        for (int i = 0; i < tapCount; i++) {
            shift(taps[i], i);
            taps[i] *= factor;
            taps[i] /= total;
        }
    }

    float _cutoff = 0.0f, _transWidth = 1.0f, _sampleRate = 1.0f;
};

class BlackmanWindow : public blackman_params {
public:
    BlackmanWindow() {}
    BlackmanWindow(float cutoff, float transWidth, float sampleRate) { init(cutoff, transWidth, sampleRate); }
    void init(float cutoff, float transWidth, float sampleRate) { keep(cutoff, transWidth, sampleRate); }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        design(taps, tapCount, factor, [](float&, int) {});
    }
};

class BlackmanBandpassWindow : public blackman_params {
public:
    BlackmanBandpassWindow() {}
    BlackmanBandpassWindow(float cutoff, float transWidth, float offset, float sampleRate) { init(cutoff, transWidth, offset, sampleRate); }
    void init(float cutoff, float transWidth, float offset, float sampleRate) {
        keep(cutoff, transWidth, sampleRate);
        _offset = offset;
    }
    void setOffset(float offset) { _offset = offset; }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        // the low-pass prototype moved to the band centre by a cosine at offset/sampleRate cycles per sample
        design(taps, tapCount, factor, [this](float& tap, int i) { tap *= cosf(2.0f * (_offset / _sampleRate) * FL_M_PI * (float)i); });
    }

private:
    float _offset = 0.0f;
};

}  // namespace filter_window

// Root-raised-cosine taps (the reference credits GNU Radio's firdes::root_raised_cosine for the closed form).
// The arithmetic is double throughout, fed by the FLOAT constant FL_M_PI and a float samples-per-symbol
// ratio, and every expression keeps the reference's operand order (src/dsp/window.h:181-225), because the
// taps must come out bit for bit.  Three cases per tap, by t = offset from the centre tap and
// q = (4 alpha t / sps)^2 - 1:
//   regular (|q| >= 1e-6): 4 alpha (cos((1+alpha) pi t/sps) + S) / (q pi), S = sin((1-alpha) pi t/sps) / (4 alpha t/sps),
//                          or its limit (1-alpha) pi / (4 alpha) at the centre tap;
//   singular, alpha == 1:  the tap is -1;
//   singular otherwise:    the l'Hopital form below.
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
    void setTapCount(int count) { _tapCount = count; }
    void setSampleRate(float sampleRate) { _sampleRate = sampleRate; }
    void setBaudRate(float baudRate) { _baudRate = baudRate; }
    void setAlpha(float alpha) { _alpha = alpha; }

    void createTaps(float* taps, int tapCount, float factor = 1.0f) override {
        (void)factor;                       // ignored, as in the reference
        tapCount |= 1;                      // (sic) an even request writes one tap more
        const int centre = tapCount / 2;
        const double sps = _sampleRate / _baudRate;
        double total = 0;
        for (int i = 0; i < tapCount; i++) {
            taps[i] = (float)tapAt(i, centre, sps);
            total += taps[i];
        }
        for (int i = 0; i < tapCount; i++) { taps[i] = taps[i] / total; }
    }

private:
    double tapAt(int i, int centre, double sps) const {
        const double t = i - centre;
        const double arg = FL_M_PI * t / sps;
        const double edge = 4 * _alpha * t / sps;
        const double q = edge * edge - 1;
        if (fabs(q) >= 0.000001) {
            const double tail = (i != centre) ? sin((1 - _alpha) * arg) / (4 * _alpha * t / sps) : (1 - _alpha) * FL_M_PI / (4 * _alpha);
            const double num = cos((1 + _alpha) * arg) + tail;
            return 4 * _alpha * num / (q * FL_M_PI);
        }
        if (_alpha == 1) { return -1; }
        const double lo = (1 - _alpha) * arg, hi = (1 + _alpha) * arg;
        const double num = (sin(hi) * (1 + _alpha) * FL_M_PI - cos(lo) * ((1 - _alpha) * FL_M_PI * sps) / (4 * _alpha * t) +
                            sin(lo) * sps * sps / (4 * _alpha * t * t));
        return 4 * _alpha * num / (-32 * FL_M_PI * _alpha * _alpha * t / sps);
    }

    int _tapCount = 0;
    float _sampleRate = 1.0f, _baudRate = 1.0f, _alpha = 0.35f;
};

}  // namespace dsp
