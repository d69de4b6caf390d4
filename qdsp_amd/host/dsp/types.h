// dsp/types.h -- sample types of the block graph.
//
// Layout contract (reference: src/dsp/types.h:7-67, :69-84): complex_t is a POD pair of
// floats {re, im}, 8 bytes, interleaved -- it is what the HIP kernels read and write as
// float2 and what the reference casts to lv_32fc_t* (src/dsp/filter.h:65).  stereo_t is
// the same shape with {l, r}.  FL_M_PI is the reference's float pi (types.h:4); every
// frequency -> phase-increment computation must use it, not M_PI, for parity.
#pragma once
#include <cmath>
#include <type_traits>

#define FL_M_PI 3.1415926535f

namespace dsp {

struct complex_t {
    float re;
    float im;

    complex_t operator*(const float s) const { return {re * s, im * s}; }
    complex_t operator/(const float s) const { return {re / s, im / s}; }
    complex_t operator*(const complex_t& o) const { return {re * o.re - im * o.im, im * o.re + re * o.im}; }
    complex_t operator+(const complex_t& o) const { return {re + o.re, im + o.im}; }
    complex_t operator-(const complex_t& o) const { return {re - o.re, im - o.im}; }

    complex_t conj() const { return {re, -im}; }
    float phase() const { return atan2f(im, re); }
    float amplitude() const { return sqrtf(re * re + im * im); }

    // Octant approximation of atan2 (reference types.h:34-51): same piecewise-linear map.
    float fastPhase() const {
        if (re == 0.0f && im == 0.0f) { return 0.0f; }
        const float a = fabsf(im);
        const float q = FL_M_PI / 4.0f;
        const float ang = (re >= 0.0f) ? q - q * ((re - a) / (re + a)) : 3.0f * q - q * ((re + a) / (a - re));
        return im < 0.0f ? -ang : ang;
    }

    // alpha-max-plus-beta-min.  The reference (types.h:57-62) takes |re| for BOTH operands,
    // which makes it |re| * 1.4; kept as is so graphs that use it see the same numbers.
    float fastAmplitude() const {
        const float a = fabsf(re), b = fabsf(re);
        return a > b ? a + 0.4f * b : b + 0.4f * a;
    }
};

struct stereo_t {
    float l;
    float r;

    stereo_t operator*(const float s) const { return {l * s, r * s}; }
    stereo_t operator+(const stereo_t& o) const { return {l + o.l, r + o.r}; }
    stereo_t operator-(const stereo_t& o) const { return {l - o.l, r - o.r}; }
};

static_assert(sizeof(complex_t) == 8 && std::is_trivially_copyable<complex_t>::value, "complex_t must stay a float pair");
static_assert(sizeof(stereo_t) == 8 && std::is_trivially_copyable<stereo_t>::value, "stereo_t must stay a float pair");

}  // namespace dsp
