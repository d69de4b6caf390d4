// cpk.hip.h -- complex arithmetic on packed FP32 pairs (gfx950 v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32: one VALU issue does the real and the imaginary lane).  A complex number is a
// 2-vector {re, im}; hipcc selects the packed instructions for ext_vector_type(2) float
// arithmetic and folds lane swaps (.yx, .xx, .yy) and whole-vector negation into
// op_sel / neg modifiers.  What it does NOT fold is negating ONE lane, so every product
// here is phrased so that the one-lane sign sits in a constant (or precomputed) operand:
//     a * b = a.xx * b + a.yy * bj,      bj = j*b = {-b.y, b.x}
// and rotations by +-j are an fma with the constant {1,-1} / {-1,1} on the swapped operand.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f mk2(float x, float y) { return (v2f){x, y}; }
// j*b: the companion operand of a product (precompute it for operands that are reused)
__device__ __forceinline__ v2f jtimes(v2f b) { return (v2f){-b.y, b.x}; }
// a * b with bj = jtimes(b)            (2 packed instructions)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b, v2f bj) { return pk_fma(a.xx, b, a.yy * bj); }
// a * b + c                            (2 packed instructions)
__device__ __forceinline__ v2f pk_cmac(v2f a, v2f b, v2f bj, v2f c) { return pk_fma(a.yy, bj, pk_fma(a.xx, b, c)); }
// t + (-j) d (forward) / t + (+j) d (inverse), and the opposite sign
template <bool INV> __device__ __forceinline__ v2f pk_add_mulj(v2f t, v2f d) {
    return pk_fma(d.yx, INV ? mk2(-1.0f, 1.0f) : mk2(1.0f, -1.0f), t);
}
template <bool INV> __device__ __forceinline__ v2f pk_sub_mulj(v2f t, v2f d) {
    return pk_fma(d.yx, INV ? mk2(1.0f, -1.0f) : mk2(-1.0f, 1.0f), t);
}
// multiply by -j (forward) / +j (inverse)
template <bool INV> __device__ __forceinline__ v2f pk_mulj(v2f a) {
    return a.yx * (INV ? mk2(-1.0f, 1.0f) : mk2(1.0f, -1.0f));
}

// a * b (CONJ: a * conj(b)) when no companion of b is at hand: 3 packed instructions
//   a b      = b.xx a + b.yy (j a),      j a  = a.yx * {-1, 1}
//   a conj b = b.xx a + b.yy (-j a),    -j a  = a.yx * { 1,-1}
template <bool CONJ> __device__ __forceinline__ v2f pk_cmulc(v2f a, v2f b) {
    return pk_fma(b.yy * a.yx, CONJ ? mk2(1.0f, -1.0f) : mk2(-1.0f, 1.0f), b.xx * a);
}

// a * b (CONJ: a * conj(b)) in TWO packed instructions and without a companion operand: v_pk_mul_f32 + v_pk_fma_f32 with the lane
// swap on op_sel and the ONE-lane sign on neg_lo / neg_hi -- modifiers the ISA has but hipcc does not select (it folds
// whole-vector negation only), hence inline asm.  (Round 3; fir_fft_dmapk_kernel, chan_uniform_kernel.)
template <bool CONJ> __device__ __forceinline__ v2f pk_cmul2(v2f a, v2f b) {   // a * b, CONJ: a * conj(b)
    v2f r;
    if (!CONJ)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "=&v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(r) : "v"(a), "v"(b));
    return r;
}

// a * b + c, companion-free (two packed FMAs)
__device__ __forceinline__ v2f pk_cmac2(v2f a, v2f b, v2f c) {
    v2f r = c;
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "+&v"(r) : "v"(a), "v"(b));      // early clobber: the first FMA writes r while a and b are still to be read by the second
    return r;
}

template <bool INV> __device__ __forceinline__ void pk_fft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = pk_add_mulj<INV>(t1, d);
    a3 = pk_sub_mulj<INV>(t1, d);
}

// v * w for a compile-time constant w = (wr, wi) of the FORWARD transform; the inverse uses conj(w)
template <bool INV> __device__ __forceinline__ v2f pk_cmul_const(v2f v, float wr, float wi) {
    const float s = INV ? -wi : wi;
    return pk_fma(v.xx, mk2(wr, s), v.yy * mk2(-s, wr));
}

// In-register 16-point DFT, radix 4x4.  Input v[n]; output X[k] is left at v[rev16(k)] (cfft.hip.h).
template <bool INV> __device__ __forceinline__ void pk_fft16(v2f (&v)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
#pragma unroll
    for (int n0 = 0; n0 < 4; n0++) pk_fft4<INV>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    // v[4*k0 + n0] *= W16^(n0*k0); forward W = exp(-j 2pi/16)
    v[4 * 1 + 1] = pk_cmul_const<INV>(v[4 * 1 + 1], c1, -s1);
    v[4 * 1 + 2] = pk_cmul_const<INV>(v[4 * 1 + 2], r, -r);
    v[4 * 1 + 3] = pk_cmul_const<INV>(v[4 * 1 + 3], s1, -c1);
    v[4 * 2 + 1] = pk_cmul_const<INV>(v[4 * 2 + 1], r, -r);
    v[4 * 2 + 2] = pk_mulj<INV>(v[4 * 2 + 2]);
    v[4 * 2 + 3] = pk_cmul_const<INV>(v[4 * 2 + 3], -r, -r);
    v[4 * 3 + 1] = pk_cmul_const<INV>(v[4 * 3 + 1], s1, -c1);
    v[4 * 3 + 2] = pk_cmul_const<INV>(v[4 * 3 + 2], -r, -r);
    v[4 * 3 + 3] = pk_cmul_const<INV>(v[4 * 3 + 3], -c1, s1);
#pragma unroll
    for (int k0 = 0; k0 < 4; k0++) pk_fft4<INV>(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}

}  // namespace qk
