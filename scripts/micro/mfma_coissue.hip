// mfma_coissue.hip -- what one v_mfma_f32_16x16x4_f32 costs a wave's VALU stream on gfx950.
//
// Question: can a radix-16 DFT pass of fir_fft_kernel move to the otherwise idle FP32 matrix
// pipe "for free"?  A 16-point DFT over 64 columns is 4 tiles x 16 MFMAs (4 real products x 4 k-steps) = 64 MFMAs per
// wave and pass, against ~150 VALU instructions (mostly two-operand adds) for the radix-4x4 butterfly it replaces.
// This benchmark measures, at 1 / 2 / 4 waves per SIMD:
//   adds        64 independent v_add_f32 per iteration
//   adds+8      the same 64 adds with one MFMA after every 8th add   (8 MFMAs per iteration)
//   adds+16     one MFMA after every 4th add                          (16 MFMAs per iteration)
//   adds+32     one MFMA after every 2nd add                          (32 MFMAs per iteration)
//   mfma        16 MFMAs per iteration, no adds
// The extra time per MFMA in the mixed rows is what the matrix instruction takes out of the wave's / SIMD's issue slots.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4f __attribute__((ext_vector_type(4)));

#define ADD(i) "v_add_f32 %" #i ", %" #i ", %16\n\t"
#define MF(j) "v_mfma_f32_16x16x4_f32 %" #j ", %21, %22, %" #j "\n\t"

// operands: %0..%15 = a[0..15], %16 = s, %17..%20 = four accumulators, %21 / %22 = the A / B operand
template <int MODE> __global__ void k(float* out, int iters, float s) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float ma = 1.0f + threadIdx.x * 1e-3f, mb = 0.5f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15)
                         ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15)
                         ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15)
                         ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15)
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                         : "v"(s));
        } else if (MODE == 1) {   // 8 MFMAs: one per 8 adds
#define G8(j) ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) MF(j)
#define H8(j) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15) MF(j)
            asm volatile(G8(17) H8(18) G8(19) H8(20) G8(17) H8(18) G8(19) H8(20)
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(s), "+v"(c0),
                           "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(ma), "v"(mb));
        } else if (MODE == 2) {   // 16 MFMAs: one per 4 adds
#define G4(x, y, z, w, j) ADD(x) ADD(y) ADD(z) ADD(w) MF(j)
#define Q16 G4(0, 1, 2, 3, 17) G4(4, 5, 6, 7, 18) G4(8, 9, 10, 11, 19) G4(12, 13, 14, 15, 20)
            asm volatile(Q16 Q16 Q16 Q16
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(s), "+v"(c0),
                           "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(ma), "v"(mb));
        } else if (MODE == 3) {   // 32 MFMAs: one per 2 adds
#define G2(x, y, j) ADD(x) ADD(y) MF(j)
#define Q8 G2(0, 1, 17) G2(2, 3, 18) G2(4, 5, 19) G2(6, 7, 20) G2(8, 9, 17) G2(10, 11, 18) G2(12, 13, 19) G2(14, 15, 20)
            asm volatile(Q8 Q8 Q8 Q8
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(s), "+v"(c0),
                           "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(ma), "v"(mb));
        } else {                  // 16 MFMAs, no adds
#define M4 MF(17) MF(18) MF(19) MF(20)
            asm volatile(M4 M4 M4 M4
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(s), "+v"(c0),
                           "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(ma), "v"(mb));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i];
    r += c0.x + c1.y + c2.z + c3.w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* d, int adds, int mfmas) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int iters = 4000, grid = 256 * 4 * wps;   // wps one-wave blocks per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // nominal cycles (2.4 GHz) one SIMD spends per iteration of ONE wave's stream
        const double cyc = ms * 1e-3 * 2.4e9 / iters / wps;
        printf("%-10s %d adds + %2d mfma per iteration, %d waves/SIMD: %.3f ms -> %.0f cyc per wave-iteration per SIMD\n", name, adds,
               mfmas, wps, ms, cyc);
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 4 * 64 * sizeof(float));
    run<0>("adds", d, 64, 0);
    run<1>("adds+8", d, 64, 8);
    run<2>("adds+16", d, 64, 16);
    run<3>("adds+32", d, 64, 32);
    run<4>("mfma", d, 0, 16);
    return 0;
}
